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
            # (fold + tap) * lrelu'(mask): the tap is the gradient of another reader of the ACTIVATED input
            taps = [torch.randn_like(it[3]) for it in items]
            plain = conv.reflect_fold_group([it[:3] for it in items], 1, 1.0)
            got = conv.reflect_fold_group([it + (t,) for it, t in zip(items, taps)], 1, 0.2, tap_first=True)
            for o, f, t, it in zip(got, plain, taps, items):
                want = (f.float() + t.float()) * torch.where(it[3].float() > 0, 1.0, 0.2)
                torch.testing.assert_close(o.float(), want, rtol=1e-2 if dt == torch.bfloat16 else 1e-6, atol=1e-2 if dt == torch.bfloat16 else 1e-6)
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
        # grouped call: the wave-split members share ONE grid per configuration (the phases of a transposed convolution,
        # of a strided layer's data gradient), results of single launches bit for bit
        from msmctts_amd.hip import lib
        torch.manual_seed(0)
        x = torch.randn(2, 1, 70, 512).bfloat16()
        items = []
        for k, Co in ((3, 64), (3, 72), (3, 96)):
            geom = conv.Geometry(1, 70, (1, k), (1, 1), (1, 1), (0, k // 2), False)
            items.append(dict(x=x, w=(torch.randn(k, Co, 512) / (512 * k) ** 0.5).bfloat16(), geom=geom, bias=torch.randn(Co),
                              in_slope=0.1))
        conv._PLANS.clear()
        singles = [conv.conv_forward(**it) for it in items]
        n0 = lib.get().msmc_conv_launch_count()
        grouped = conv.conv_forward_group(items)
        assert lib.get().msmc_conv_launch_count() - n0 == 1
        assert b'conv_gather_ks_group_kernel' in lib.get().msmc_conv_last_kernel()
        for a, b in zip(grouped, singles):
            assert torch.equal(a, b)
    finally:
        conv._build_desc = real
        conv._GATHER_CANDIDATES = saved
    assert len(used) >= 6


def test_vq_edge_cases_empty_single_frame_zero_length_ragged():
    _parity.check_vq_edge_cases('cpu')


@pytest.mark.parametrize('variant', [16, 17, 18, 19, 20, 21, 24, 25, 26, 27, 28, 29, 31])        # (22, 23, 30: retired in round 6)
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


@pytest.mark.parametrize('variant', [32, 33])
def test_gather_persistent_thin_layer_variant(variant):
    """variant 32 / 33 = the same with the epilogue of a tile deferred into the next iteration (gather4.inc: weights of all taps resident in LDS, halo tiles of consecutive pixel tiles by LDS-DMA,
    accumulators [co][pixel] with the epilogue in registers) on the thin-layer family it serves: forward and data
    gradient (mask operand), residual operands / output division / output leaky-ReLU against the second generation,
    more tiles than workgroups, ragged last tile, taps along H, layers outside its scope refused"""
    from msmctts_amd.hip import conv, lib
    cases = [('g4 k3 64->64', 2, 64, 64, 1, 300, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
             ('g4 k11 d5 32->32', 1, 32, 32, 1, 200, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1),
             ('g4 k7 d3 64->32', 1, 64, 32, 1, 130, (1, 7), (1, 1), (1, 3), (0, 9), False, 1.0),
             ('g4 k5x1 32->64 p3', 2, 32, 64, 22, 3, (5, 1), (1, 1), (1, 1), (2, 0), False, 0.2),
             ('g4 valid k3 64->64', 1, 64, 64, 1, 37, (1, 3), (1, 1), (1, 1), (0, 0), False, 1.0),
             ('g4 tiny L3 k11 d5', 1, 32, 32, 1, 3, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1)]
    real = conv._build_desc
    state = {'variant': variant}

    def forced(*a, **k):
        d = real(*a, **k)
        if d.dtype == 1:
            d.variant = state['variant']
        return d
    conv._build_desc = forced
    try:
        for grid in (0, 2):                          # 2: six tiles on two workgroups -- the ring wraps
            lib.get().msmc_conv_set_gather4_grid(grid)
            for case in cases[:2] if grid else cases:
                for part in ('fwd', 'dgrad'):
                    conv._PLANS.clear()
                    _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', parts=(part,))
                    assert b'conv_gather4' in lib.get().msmc_conv_last_kernel(), (case[0], part)
        lib.get().msmc_conv_set_gather4_grid(0)
        # epilogue operands against the second generation on the same inputs
        torch.manual_seed(0)
        B, C, Lx, k, dil = 2, 64, 300, 7, 3
        geom = conv.Geometry(1, Lx, (1, k), (1, 1), (1, dil), (0, dil * (k - 1) // 2), False)
        x = torch.randn(B, 1, Lx, C).bfloat16()
        w = (torch.randn(k, C, C) / (C * k) ** 0.5).bfloat16()
        bias, res, res2 = torch.randn(C), torch.randn(B, 1, Lx, C).bfloat16(), torch.randn(B, 1, Lx, C).bfloat16()
        outs = []
        for v in (variant, 2):
            state['variant'] = v
            conv._PLANS.clear()
            outs.append(conv.conv_forward(x, w, geom, bias=bias, in_slope=0.1, res=res, res2=res2, out_div=3.0,
                                          out_slope=0.2))
        assert _convcases.rel(outs[0], outs[1]) < 1e-2
        # grouped: three members on one persistent grid (msmc_conv_set_gather4_grouping(1)), six + six + three tiles on
        # a capped number of workgroups, equal to one launch per member
        if variant == 32:
            state['variant'] = 32
            items, singles = [], []
            for kk, dd in ((3, 1), (7, 3), (11, 1)):
                gm = conv.Geometry(1, Lx, (1, kk), (1, 1), (1, dd), (0, dd * (kk - 1) // 2), False)
                ww = (torch.randn(kk, C, C) / (C * kk) ** 0.5).bfloat16()
                items.append(dict(x=x, w=ww, geom=gm, bias=bias, in_slope=0.1, res=res))
                conv._PLANS.clear()
                singles.append(conv.conv_forward(x, ww, gm, bias=bias, in_slope=0.1, res=res))
            lib.get().msmc_conv_set_gather4_grouping(1)
            try:
                for grid in (0, 2):
                    lib.get().msmc_conv_set_gather4_grid(grid)
                    conv._PLANS.clear()
                    outs_g = conv.conv_forward_group(items)
                    assert b'conv_gather4_group_kernel' in lib.get().msmc_conv_last_kernel()
                    assert all(torch.equal(a, b) for a, b in zip(outs_g, singles))
            finally:
                lib.get().msmc_conv_set_gather4_grouping(0)
                lib.get().msmc_conv_set_gather4_grid(0)
        # outside the scope: stride 3 / 96 channels -> MSMC_E_SHAPE surfaces as an error, nothing is mis-computed
        state['variant'] = variant
        conv._PLANS.clear()
        with pytest.raises(RuntimeError, match='msmc_conv_gather'):
            _convcases.check_conv_case(_convcases.SMALL[2], torch.bfloat16, 2e-2, 'cpu', parts=('fwd',))
    finally:
        conv._build_desc = real
        conv._PLANS.clear()
        lib.get().msmc_conv_set_gather4_grid(0)


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


WG4_CASES = [
    # (name, B, Cin, Cout, H, W, kernel, stride, dilation, padding, reflect, in_slope)
    ('ffn k3 128->64', 2, 128, 64, 1, 100, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
    ('gen k11 d5 C64', 2, 64, 64, 1, 70, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1),
    ('gen k7 d3 C64 long', 1, 64, 64, 1, 400, (1, 7), (1, 1), (1, 3), (0, 9), False, 0.1),
    ('mpd 64->128 p3 s1', 2, 64, 128, 22, 3, (5, 1), (1, 1), (1, 1), (2, 0), False, 0.2),
    ('valid k3 64->64', 1, 64, 64, 1, 37, (1, 3), (1, 1), (1, 1), (0, 0), False, 1.0),
    ('wide pad k3 64->64', 1, 64, 64, 1, 20, (1, 3), (1, 1), (1, 1), (0, 2), False, 1.0),
    ('tiny L3 k11 d5 C64', 1, 64, 64, 1, 3, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1),
    ('ffn k3 88->200 (channel counts that end inside a tile of 64)', 2, 88, 200, 1, 70, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.0),
    ('k3 136->72 lrelu', 1, 136, 72, 1, 45, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
]


@pytest.mark.parametrize('gen', [4, 5, 6])
def test_wgrad_fourth_generation_ring(gen):
    """generation 4 of the bf16 weight gradient (wgrad4.inc: LDS-DMA ring of 3 / 2 / 4 pixel-tile stages, source-side
    swizzle, flattened pixel axis) against PyTorch on the interpreter: model split, one split (direct accumulation, the
    ring wraps) and forced splits; shapes outside its scope fall back to the third generation"""
    from msmctts_amd.hip import conv, lib
    L = lib.get()
    L.msmc_conv_set_wgrad_generation(gen)
    try:
        for split in (0, 1, 3):
            L.msmc_conv_set_wgrad_split(split)
            for case in WG4_CASES:
                conv._PLANS.clear()
                _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', parts=('wgrad',))
        L.msmc_conv_set_wgrad_split(0)
        # the kernel that ran last for an in-scope layer is the fourth generation; out of scope (stride 3) it is not
        name, B, Cin, Cout, H, W, k, s, dil, pad, reflect, slope = WG4_CASES[0]
        geom = conv.Geometry(H, W, k, s, dil, pad, reflect)
        x, g = torch.randn(B, H, W, Cin).bfloat16(), torch.randn(B, geom.Hout, geom.Wout, Cout).bfloat16()
        conv.conv_wgrad(x, g, geom, k[0] * k[1], db=torch.zeros(Cout))
        assert b'conv_wgrad4_kernel' in L.msmc_conv_last_kernel()
        x, g = torch.randn(2, 1, 70, 88).bfloat16(), torch.randn(2, 1, 70, 200).bfloat16()     # (600 / 1536-wide FFT blocks: tiles of 64 that end past the channels)
        conv.conv_wgrad(x, g, conv.Geometry(1, 70, (1, 3), (1, 1), (1, 1), (0, 1), False), 3, db=torch.zeros(200))
        assert b'conv_wgrad4_kernel' in L.msmc_conv_last_kernel()
        conv._PLANS.clear()
        _convcases.check_conv_case(_convcases.SMALL[5], torch.bfloat16, 2e-2, 'cpu', parts=('wgrad',))
        assert b'conv_wgrad4_kernel' not in L.msmc_conv_last_kernel()
    finally:
        L.msmc_conv_set_wgrad_generation(2)
        L.msmc_conv_set_wgrad_split(0)


WG7_CASES = [
    ('w7 ffn k3 136->200 (both channel counts end inside the second tile half)', 2, 136, 200, 1, 70, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.0),
    ('w7 k3 256->128 lrelu', 1, 256, 128, 1, 45, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
    ('w7 k5 d2 128->264: two tap groups', 2, 128, 264, 1, 75, (1, 5), (1, 1), (1, 2), (0, 4), False, 1.0),
    ('w7 1-tap 384->128', 2, 384, 128, 1, 100, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0),
    ('w7 k5x1 128->128 over [H][W]', 2, 128, 128, 22, 3, (5, 1), (1, 1), (1, 1), (2, 0), False, 0.2),
    ('w7 tiny L3 k3 128->128', 1, 128, 128, 1, 3, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
]


def test_wgrad_128_channel_tiles():
    """variant 9 (wgrad7.inc: the fourth generation's ring and pixel geometry on 128 x 128 channel tiles, eight waves of 64 x 32 x taps)
    against PyTorch on the interpreter: model split, one split (direct accumulation, the ring wraps) and forced splits; channel
    counts that end inside a tile, two tap groups, one tap, k x 1 taps over images; layers under 128 channels are refused"""
    from msmctts_amd.hip import conv, lib
    L = lib.get()
    real = conv._build_desc

    def forced(*a, **k):
        d = real(*a, **k)
        if d.dtype == 1:
            d.variant = 9
        return d
    conv._build_desc = forced
    try:
        for split in (0, 1, 3):
            L.msmc_conv_set_wgrad_split(split)
            for case in WG7_CASES:
                conv._PLANS.clear()
                _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', parts=('wgrad',))
                assert b'conv_wgrad7_kernel' in L.msmc_conv_last_kernel() or b'reduce' in L.msmc_conv_last_kernel(), (case[0], L.msmc_conv_last_kernel())
        L.msmc_conv_set_wgrad_split(0)
        conv._PLANS.clear()
        with pytest.raises(RuntimeError):
            _convcases.check_conv_case(('thin', 1, 64, 128, 1, 40, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0), torch.bfloat16, 2e-2, 'cpu', parts=('wgrad',))
    finally:
        conv._build_desc = real
        L.msmc_conv_set_wgrad_split(0)


def test_wgrad_general_lattice_dma_staging():
    """variant 7 (wgrad5.inc: the third generation's lattice tiles and table-driven fragment rows with both operands
    staged by LDS-DMA into a two-stage ring) on strided, 2-D, reflection-padded and dilated layers, model split, one
    split (the ring wraps) and forced splits, against PyTorch on the interpreter"""
    from msmctts_amd.hip import conv, lib
    L = lib.get()
    cases = [('w5 mpd 64->128 s3 p3', 2, 64, 128, 40, 3, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
             ('w5 mrd 64->64 s2 reflect', 1, 64, 64, 13, 18, (3, 3), (2, 2), (1, 1), (1, 1), True, 1.0),
             ('w5 mrd 64->128 s1x2 reflect', 1, 64, 128, 9, 20, (3, 3), (1, 2), (1, 1), (1, 1), True, 0.2),
             ('w5 gen k11 d5 C64', 2, 64, 64, 1, 70, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1),
             ('w5 tiny 3x2 reflect s2', 2, 64, 64, 3, 2, (3, 3), (2, 2), (1, 1), (1, 1), True, 1.0)]
    L.msmc_conv_set_wgrad_generation(7)
    try:
        for split in (0, 1, 3):
            L.msmc_conv_set_wgrad_split(split)
            for case in cases:
                conv._PLANS.clear()
                _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', parts=('wgrad',))
        L.msmc_conv_set_wgrad_split(0)
        name, B, Cin, Cout, H, W, k, s, dil, pad, reflect, slope = cases[0]
        geom = conv.Geometry(H, W, k, s, dil, pad, reflect)
        x, g = torch.randn(B, H, W, Cin).bfloat16(), torch.randn(B, geom.Hout, geom.Wout, Cout).bfloat16()
        conv._PLANS.clear()
        conv.conv_wgrad(x, g, geom, k[0] * k[1], db=torch.zeros(Cout))
        assert b'conv_wgrad5_kernel' in L.msmc_conv_last_kernel()
        # grouped: two strided members of different tap counts on one grid, a third member the old way; bit-identical to
        # one launch per member at the same split
        import ctypes
        L.msmc_conv_set_wgrad_generation(2)
        L.msmc_conv_set_wgrad_split(2)
        torch.manual_seed(0)
        descs, gs, refs, outs = [], [], [], []
        for (H, W, k, s, pad, reflect, cout, variant) in ((40, 3, (5, 1), (3, 1), (2, 0), False, 64, 7),
                                                          (13, 18, (3, 3), (2, 2), (1, 1), True, 128, 7),
                                                          (9, 20, (3, 3), (1, 2), (1, 1), True, 64, 3)):
            geom = conv.Geometry(H, W, k, s, (1, 1), pad, reflect)
            x = torch.randn(2, H, W, 64).bfloat16()
            g = torch.randn(2, geom.Hout, geom.Wout, cout).bfloat16()
            d = conv._build_desc(x.dtype, 2, H, W, 64, geom.Hout, geom.Wout, cout, geom.fwd_lattice, geom.fwd_taps,
                                 1 if reflect else 0, 0.2, 1.0, 1.0, 1.0)
            d.x = d.w = d.out = x.data_ptr()
            d.variant, d.dw_copies = variant, 1
            T = k[0] * k[1]
            need = L.msmc_conv_wgrad_workspace(ctypes.byref(d), g.data_ptr())
            ws = torch.zeros(max(1, need // 4))
            dw_ref, db_ref = torch.zeros(T, cout, 64), torch.zeros(cout)
            assert L.msmc_conv_wgrad_ws(ctypes.byref(d), g.data_ptr(), dw_ref.data_ptr(), db_ref.data_ptr(), ws.data_ptr(),
                                        need, None) == 0
            descs.append(d); gs.append((x, g)); refs.append((dw_ref, db_ref))
            outs.append((torch.zeros(T, cout, 64), torch.zeros(cout)))
        arr = (lib.ConvDesc * 3)(*descs)
        vp = ctypes.c_void_p * 3
        need = sum(L.msmc_conv_wgrad_workspace(ctypes.byref(d), xg[1].data_ptr()) for d, xg in zip(descs, gs))
        ws = torch.zeros(max(1, need // 4))
        rc = L.msmc_conv_wgrad_group_ws4(arr, vp(*[xg[1].data_ptr() for xg in gs]), vp(*[o[0].data_ptr() for o in outs]),
                                         vp(*[o[1].data_ptr() for o in outs]), 3, ws.data_ptr(), need, None, 1)
        assert rc == 0, rc
        for (dw_ref, db_ref), (dw, db) in zip(refs, outs):
            assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
        arr2 = (lib.ConvDesc * 2)(*descs[:2])
        vp2 = ctypes.c_void_p * 2
        rc = L.msmc_conv_wgrad_group_ws4(arr2, vp2(*[xg[1].data_ptr() for xg in gs[:2]]), vp2(*[o[0].data_ptr() for o in outs[:2]]),
                                         vp2(*[o[1].data_ptr() for o in outs[:2]]), 2, ws.data_ptr(), need, None, 1)
        assert rc == 0 and b'conv_wgrad5_group_kernel' in L.msmc_conv_last_kernel(), L.msmc_conv_last_kernel()
    finally:
        L.msmc_conv_set_wgrad_generation(2)
        L.msmc_conv_set_wgrad_split(0)


def test_wgrad_fourth_generation_grouped():
    """msmc_conv_wgrad_group_ws4(group4 = 1): the fourth-generation members of a grouped call share one grid of their own
    kernel (members of different tap counts: the widest sets the accumulator budget), a member outside the scope goes the
    old way; bit-identical to one launch per member with the same split"""
    import ctypes
    from msmctts_amd.hip import conv, lib
    L = lib.get()
    torch.manual_seed(0)
    B, C, Lx = 2, 64, 90
    x = torch.randn(B, 1, Lx, C).bfloat16()
    for split in (1, 2):
        L.msmc_conv_set_wgrad_split(split)
        descs, gs, refs, outs = [], [], [], []
        for k, dil, cout, variant in ((3, 1, 64, 4), (7, 3, 128, 4), (11, 1, 64, 5), (3, 1, 40, 4)):
            geom = conv.Geometry(1, Lx, (1, k), (1, 1), (1, dil), (0, dil * (k - 1) // 2), False)
            g = torch.randn(B, 1, Lx, cout).bfloat16()
            d = conv._build_desc(x.dtype, B, 1, Lx, C, 1, Lx, cout, geom.fwd_lattice, geom.fwd_taps, 0, 0.1, 1.0, 1.0, 1.0)
            d.x = d.w = d.out = x.data_ptr()
            d.variant, d.dw_copies = variant, 1
            if cout == 40:
                d.variant = 3                     # outside the fourth generation's scope (Cout % 64)
            need = L.msmc_conv_wgrad_workspace(ctypes.byref(d), g.data_ptr())
            ws = torch.zeros(max(1, need // 4))
            dw_ref, db_ref = torch.zeros(k, cout, C), torch.zeros(cout)
            assert L.msmc_conv_wgrad_ws(ctypes.byref(d), g.data_ptr(), dw_ref.data_ptr(), db_ref.data_ptr(), ws.data_ptr(),
                                        need, None) == 0
            descs.append(d); gs.append(g); refs.append((dw_ref, db_ref))
            outs.append((torch.zeros(k, cout, C), torch.zeros(cout)))
        n = len(descs)
        arr = (lib.ConvDesc * n)(*descs)
        vp = ctypes.c_void_p * n
        need = sum(L.msmc_conv_wgrad_workspace(ctypes.byref(d), g.data_ptr()) for d, g in zip(descs, gs))
        ws = torch.zeros(max(1, need // 4))
        rc = L.msmc_conv_wgrad_group_ws4(arr, vp(*[g.data_ptr() for g in gs]), vp(*[o[0].data_ptr() for o in outs]),
                                         vp(*[o[1].data_ptr() for o in outs]), n, ws.data_ptr(), need, None, 1)
        assert rc == 0, rc
        for (dw_ref, db_ref), (dw, db) in zip(refs, outs):
            assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
        # the in-scope members alone: the last launch is the second stage of the shared fourth-generation grid
        arr3 = (lib.ConvDesc * 3)(*descs[:3])
        vp3 = ctypes.c_void_p * 3
        rc = L.msmc_conv_wgrad_group_ws4(arr3, vp3(*[g.data_ptr() for g in gs[:3]]), vp3(*[o[0].data_ptr() for o in outs[:3]]),
                                         vp3(*[o[1].data_ptr() for o in outs[:3]]), 3, ws.data_ptr(), need, None, 1)
        assert rc == 0 and b'conv_wgrad4_group_kernel<4, 1>' in L.msmc_conv_last_kernel(), L.msmc_conv_last_kernel()
    L.msmc_conv_set_wgrad_split(0)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_fused_add_layernorm_gate_tanh_match_torch(dtype, tol):
    _parity.check_norm_kernels('cpu', dtype, tol)


def test_hip_adamw_matches_torch_adamw_with_clipping():
    _parity.check_hip_adamw('cpu')


def test_hip_adamw_resume_then_capture_rollback_keeps_loaded_moments():
    _parity.check_hip_adamw_resume_then_capture_rollback('cpu')


def test_predictor_steps_draw_fresh_dropout_masks():
    _parity.check_predictor_dropout_masks_advance('cpu')


def test_codebook_statistics_split_update_is_the_fused_update():
    _parity.check_codebook_split_update('cpu')


def test_resblock_standalone_matches_stock_operators():
    _parity.check_resblock_standalone('cpu')


def test_window_gather_and_output_tanh_match_the_operator_chains():
    _parity.check_window_gather_and_output_tanh(torch.device('cpu'))


def test_two_autograd_graphs_over_one_bank_back_propagated_one_after_the_other():
    _parity.check_two_graphs_over_one_bank('cpu')


def test_sum_dropout_add_row_mask_kernels_and_shared_input_gradients():
    _parity.check_glue_kernels('cpu')


def test_direct_kernels_with_every_epilogue_operand_at_once():
    _parity.check_direct_kernels_with_every_epilogue_operand('cpu')


def test_output_projection_inside_the_add_layernorm_launch():
    _parity.check_fc_add_ln('cpu')


def test_front_end_chains_in_lock_step_equal_the_chains_one_by_one():
    _parity.check_fronts_lockstep(torch.device('cpu'))


def test_weight_images_in_one_tiled_pass_match_the_definition():
    _parity.check_weight_image_tiles(torch.device('cpu'))


def test_train_steps_match_reference_with_ungrouped_launches():
    """MSMC_GROUPED=0 path (one launch per convolution; tap gradients through _HipConv, incl. the reflect-padded MRD
    layers' fold + add) against the same reference fixture"""
    from msmctts_amd.hip import convnet
    keep = convnet.GROUPED
    convnet.GROUPED = False
    try:
        _parity.check_train_steps('cpu')
    finally:
        convnet.GROUPED = keep


def test_emb_autoencoder_matches_reference():
    """the QS-TTS synthesiser MSMCVQGANEmb against the reference's own module"""
    _parity.check_emb_autoencoder('cpu')


def test_inference_glue_matches_reference():
    _parity.check_inference('cpu')


def test_attention_kernels_match_the_reference_chain():
    _parity.check_attention('cpu')


@pytest.mark.parametrize('H,K,D,N', [
    (4, 64, 256, 100),     # all heads' images resident in LDS; ragged last tile
    (4, 256, 256, 70),     # one head at a time: one image buffer + one fp32 rows buffer in LDS, refilled in turns (d = 64)
    (2, 256, 128, 900),    # the same over several persistent iterations per workgroup, idle waves in the last one
    (8, 512, 256, 45),     # d = 32: one 32-wide contraction step, nine index bits
    (2, 48, 64, 33),       # codeword tiles that are not a power of two, two heads (index store of a partial head group)
    (1, 16, 64, 300),      # single head, several persistent iterations per workgroup
])
def test_vq_shortlist_search_is_bit_identical_to_the_exact_kernel_and_the_oracle(H, K, D, N):
    """csrc/vq_shortlist.inc on the interpreter: bf16 shortlist + error bound + exact re-search of unsure tiles gives the
    exact kernel's indices / quant / diff on Gaussian data, on near-ties (gap 1e-7), on duplicate codewords and on frames
    that ARE codewords; the exact path is taken for the adversarial frames and for few of the Gaussian ones."""
    import numpy as np
    from msmctts_amd.hip import vq
    from oracle import cvq
    rng = np.random.default_rng(H + K + N)
    d = D // H
    e = rng.standard_normal((H, d, K)).astype(np.float32)
    e[:, :, 7] = e[:, :, 3]                                             # exact duplicates: the first minimum is 3
    e[:, :, 11] = e[:, :, 3]                                            # ... a run of three (the two-candidate re-rank cannot settle it)
    e[:, :, 9] = e[:, :, 5] * np.float32(1 + 1e-7)                      # near-duplicate pair
    x = rng.standard_normal((N, D)).astype(np.float32)
    na = min(N // 2, 20)
    x[:na] = np.tile(e[:, :, 3].transpose(0, 1).reshape(-1), (na, 1)) + rng.standard_normal((na, D)).astype(np.float32) * 1e-4
    pp = 32 if N >= 35 else na                                          # (in a 16-frame tile of their own)
    x[pp:pp + 3] = e[:, :, 5].reshape(-1)                               # a codeword itself (and its near-duplicate)
    want = cvq.search(x, e)
    et, en = vq.vq_prepare(torch.from_numpy(e))
    assert getattr(et, 'shortlist_image', None) is not None
    vq.SLOW_COUNT = torch.zeros(2, dtype=torch.int64)
    try:
        q, df, i = vq.vq_search(torch.from_numpy(x), et, en, shortlist=True)
        slow_adv = int(vq.SLOW_COUNT.sum())
        slow_adv_full = int(vq.SLOW_COUNT[1])
        assert lib_last() == 'vq_search_sl_kernel'
        q0, df0, i0 = vq.vq_search(torch.from_numpy(x), et, en, shortlist=False)
        assert lib_last() != 'vq_search_sl_kernel'
        vq.SLOW_COUNT.zero_()
        xg = rng.standard_normal((max(N, 64), D)).astype(np.float32)
        e2 = rng.standard_normal((H, d, K)).astype(np.float32)
        et2, en2 = vq.vq_prepare(torch.from_numpy(e2))
        q2, df2, i2 = vq.vq_search(torch.from_numpy(xg), et2, en2, shortlist=True)
        slow_gauss, slow_gauss_full = int(vq.SLOW_COUNT.sum()), int(vq.SLOW_COUNT[1])
    finally:
        vq.SLOW_COUNT = None
    for got, ref, name in ((i, want['ind'], 'ind'), (q, want['quant'], 'quant'), (df, want['diff'], 'diff')):
        assert np.array_equal(got.numpy(), ref), name
    assert torch.equal(i, i0) and torch.equal(q, q0) and torch.equal(df, df0)
    w2 = cvq.search(xg, e2)
    assert np.array_equal(i2.numpy(), w2['ind']) and np.array_equal(q2.numpy(), w2['quant']) and np.array_equal(df2.numpy(), w2['diff'])
    assert slow_adv - slow_adv_full >= 1 and slow_adv_full >= 1         # the planted pairs / triples went through the two exact paths
    assert (want['ind'][:na] == 3).all() and not np.isin(want['ind'], (7, 11)).any()
    tiles = ((max(N, 64) + 15) // 16) * H
    assert slow_gauss <= max(2, tiles // 3), (slow_gauss, tiles)        # Gaussian data: mostly decided by the shortlist
    assert slow_gauss_full <= max(1, tiles // 20), (slow_gauss_full, tiles)


def lib_last():
    from msmctts_amd.hip import lib
    return lib.get().msmc_vq_last_kernel().decode()


def test_wgrad_deferred_second_stage_equals_the_immediate_one():
    """msmc_conv_wgrad_defer_* / _reduce_pending (hip/conv.py DeferredReduce): first stages of several no-atomics weight
    gradients (single, two-level, grouped) into the arena, ONE merged second stage at the end -- bit-identical to the second
    stage right behind every first stage; records are consumed, the arena is reused, nothing is deferred without a sink"""
    from msmctts_amd.hip import conv, lib
    L = lib.get()
    L.msmc_conv_set_wgrad_generation(3)
    try:
        torch.manual_seed(1)
        jobs = []
        for split, (B, C, Lx, k, dil) in ((3, (2, 64, 90, 7, 3)), (47, (1, 32, 3000, 3, 1)), (2, (2, 64, 90, 11, 1))):
            geom = conv.Geometry(1, Lx, (1, k), (1, 1), (1, dil), (0, dil * (k - 1) // 2), False)
            x, g = torch.randn(B, 1, Lx, C).bfloat16(), torch.randn(B, 1, Lx, C).bfloat16()
            jobs.append((split, geom, x, g, k, C))

        def run(defer):
            outs = []
            conv.DEFER_TO = defer
            try:
                for split, geom, x, g, k, C in jobs:
                    L.msmc_conv_set_wgrad_split(split)
                    dw, db = torch.zeros(k, C, C), torch.zeros(C)
                    conv.conv_wgrad(x, g, geom, k, in_slope=0.1, dw=dw, db=db)
                    outs.append((dw, db))
                L.msmc_conv_set_wgrad_split(2)
                items = []
                for split, geom, x, g, k, C in jobs[::2]:
                    dw, db = torch.zeros(k, C, C), torch.zeros(C)
                    items.append(dict(x=x, g=g, geom=geom, n_slices=k, in_slope=0.1, dw=dw.view(-1), db=db, copies=1))
                    outs.append((dw, db))
                conv.conv_wgrad_group(items)
            finally:
                conv.DEFER_TO = None
            return outs

        ref = run(None)
        d = conv.DeferredReduce()
        got = run(d)
        assert d.n >= 5 and any(c[1] > 0 for c in d.chunks)
        assert not all(torch.equal(a[0], b[0]) for a, b in zip(got, ref))          # (nothing added up yet)
        d.flush(lib.stream(jobs[0][2]))
        assert d.n == 0 and all(c[1] == 0 for c in d.chunks)
        for (dw, db), (dw_ref, db_ref) in zip(got, ref):
            assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
        got2 = run(d)                                                               # the arena is reused
        d.flush(lib.stream(jobs[0][2]))
        assert len(d.chunks) == 1
        for (dw, db), (dw_ref, db_ref) in zip(got2, ref):
            assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
    finally:
        L.msmc_conv_set_wgrad_generation(2)
        L.msmc_conv_set_wgrad_split(0)


@pytest.mark.parametrize('g1v', [34, 35])
def test_gather_one_tap_gemm_variant(g1v):
    """variants 34 / 35 (128 x 128 and 64 x 128 tiles; gemm1.inc: kernel-size-1 layers as a plain channel GEMM, both operands by LDS-DMA in 64-channel chunks,
    swapped operand roles with the epilogue in registers): forward and data gradient (mask operand) of the FFT-block
    projection / quantiser 1x1 shapes against PyTorch -- ragged pixel and channel tiles, a channel count that is not a
    multiple of the chunk, more pixel tiles than one XCD group, 2-D images, a contraction deeper than the four stages (ring) -- every epilogue operand against the second
    generation, layers outside its scope refused"""
    from msmctts_amd.hip import conv, lib
    cases = [('g1 qkv 256->384', 3, 256, 384, 1, 100, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0),
             ('g1 out 128->256 ragged', 2, 128, 256, 1, 77, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0),
             ('g1 in_linear 80->256', 2, 80, 256, 1, 70, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0),
             ('g1 mel 256->80 lrelu', 1, 256, 80, 1, 140, (1, 1), (1, 1), (1, 1), (0, 0), False, 0.1),
             ('g1 many tiles 64->72', 1, 64, 72, 1, 1200, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0),
             ('g1 image 96->32', 2, 96, 32, 9, 7, (1, 1), (1, 1), (1, 1), (0, 0), False, 0.2),
             ('g1 ring of stages 456->136', 1, 456, 136, 1, 90, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0)]
    real = conv._build_desc
    state = {'variant': g1v}

    def forced(*a, **k):
        d = real(*a, **k)
        if d.dtype == 1:
            d.variant = state['variant']
        return d
    conv._build_desc = forced
    try:
        for case in cases:
            for part in ('fwd', 'dgrad'):
                conv._PLANS.clear()
                _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', parts=(part,))
                assert b'conv_gemm1_kernel' in lib.get().msmc_conv_last_kernel(), (case[0], part)
        torch.manual_seed(0)
        B, Ci, Co, Lx = 2, 128, 96, 150
        geom = conv.Geometry(1, Lx, (1, 1), (1, 1), (1, 1), (0, 0), False)
        x = torch.randn(B, 1, Lx, Ci).bfloat16()
        w = (torch.randn(1, Co, Ci) / Ci ** 0.5).bfloat16()
        bias, res, res2 = torch.randn(Co), torch.randn(B, 1, Lx, Co).bfloat16(), torch.randn(B, 1, Lx, Co).bfloat16()
        outs = []
        for v in (g1v, 2):
            state['variant'] = v
            conv._PLANS.clear()
            geom = conv.Geometry(1, Lx, (1, 1), (1, 1), (1, 1), (0, 0), False)      # (a geometry keeps its descriptors)
            outs.append(conv.conv_forward(x, w, geom, bias=bias, in_slope=0.1, res=res, res2=res2, out_div=3.0, out_slope=0.2))
        assert _convcases.rel(outs[0], outs[1]) < 1e-2
        # grouped call with 1-tap members: each on a grid of its own, results of single launches
        state['variant'] = g1v
        geom = conv.Geometry(1, Lx, (1, 1), (1, 1), (1, 1), (0, 0), False)
        items = [dict(x=x, w=(torch.randn(1, Co, Ci) / Ci ** 0.5).bfloat16(), geom=geom, bias=bias, res=res) for _ in range(3)]
        conv._PLANS.clear()
        singles = [conv.conv_forward(**it) for it in items]
        for a, b in zip(conv.conv_forward_group(items), singles):
            assert torch.equal(a, b)
        # outside the scope: a 3-tap layer -> MSMC_E_SHAPE surfaces as an error, nothing is mis-computed
        conv._PLANS.clear()
        with pytest.raises(RuntimeError, match='msmc_conv_gather'):
            _convcases.check_conv_case(_convcases.SMALL[2], torch.bfloat16, 2e-2, 'cpu', parts=('fwd',))
    finally:
        conv._build_desc = real
        conv._PLANS.clear()


@pytest.mark.parametrize('variant', [40, 41, 42, 44, 45, 46, 47])        # (43: retired in round 6)
def test_gather_fifth_generation_variants(variant):
    """variants 40..44 (gather5.inc: sixteen waves, stages of (64-channel chunk, tap) with the weight slices in an LDS-DMA
    ring and the halo tile of a chunk shared by its taps, swapped operand roles, epilogue in registers with 16-byte stores):
    every tile shape where it applies, forward and data gradient -- ragged pixel and channel tiles, dilation, stride,
    reflection, 2-D taps, two taps (ring of three stages), several chunks, more pixel tiles than one XCD group -- every
    epilogue operand against the second generation, grouped calls equal to single launches, 1-tap layers refused"""
    from msmctts_amd.hip import conv, lib
    cases = [('g5 k3 64->128', 2, 64, 128, 1, 150, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
             ('g5 k11 d5 128->136', 1, 128, 136, 1, 70, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1),
             ('g5 k5x1 s3 64->256', 1, 64, 256, 40, 3, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
             ('g5 3x3 reflect s2 64->72', 1, 64, 72, 13, 18, (3, 3), (2, 2), (1, 1), (1, 1), True, 1.0),
             ('g5 k2 192->128', 1, 192, 128, 1, 45, (1, 2), (1, 1), (1, 1), (0, 1), False, 1.0),
             ('g5 k3 many tiles 64->128', 1, 64, 128, 1, 1300, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
             ('g5 ffn 256->512 relu', 2, 256, 512, 1, 100, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.0),
             ('g5 k7 d3 64->64 thin', 1, 64, 64, 1, 700, (1, 7), (1, 1), (1, 3), (0, 9), False, 0.1),
             ('g5 mrd 64->128 s2 wide halo', 2, 64, 128, 21, 40, (3, 3), (2, 2), (1, 1), (1, 1), True, 0.2),
             ('g5 k5x1 512->128 two contraction groups', 1, 512, 128, 30, 5, (5, 1), (1, 1), (1, 1), (2, 0), False, 0.2)]
    real = conv._build_desc
    state = {'variant': variant}

    def forced(*a, **k):
        d = real(*a, **k)
        if d.dtype == 1:
            d.variant = state['variant']
        return d
    conv._build_desc = forced
    ran = 0
    try:
        for case in cases:
            for part in ('fwd', 'dgrad'):
                conv._PLANS.clear()
                try:
                    _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', parts=(part,))
                    assert b'conv_gather5_kernel' in lib.get().msmc_conv_last_kernel(), (case[0], part)
                    ran += 1
                except RuntimeError as e:             # MSMC_E_SHAPE: this configuration does not apply to the layer
                    assert 'msmc_conv_gather' in str(e), e
        assert ran >= (1 if variant == 47 else 3), (variant, ran)
        if variant == 47:                  # (one-chunk wide-halo form: Cin = 64 only -- the cases above are its scope)
            return
        torch.manual_seed(0)
        B, Ci, Co, Lx = 2, 128, 264, 150
        geom = conv.Geometry(1, Lx, (1, 3), (1, 1), (1, 2), (0, 2), False)
        x = torch.randn(B, 1, Lx, Ci).bfloat16()
        w = (torch.randn(3, Co, Ci) / (3 * Ci) ** 0.5).bfloat16()
        bias, res, res2 = torch.randn(Co), torch.randn(B, 1, Lx, Co).bfloat16(), torch.randn(B, 1, Lx, Co).bfloat16()
        outs = []
        for v in (variant, 2):
            state['variant'] = v
            conv._PLANS.clear()
            fresh = conv.Geometry(1, Lx, (1, 3), (1, 1), (1, 2), (0, 2), False)         # (a geometry keeps its descriptors)
            outs.append(conv.conv_forward(x, w, fresh, bias=bias, in_slope=0.1, res=res, res2=res2, out_div=3.0, out_slope=0.2))
        assert b'conv_gather2' in lib.get().msmc_conv_last_kernel()
        assert _convcases.rel(outs[0], outs[1]) < 1e-2
        # grouped call: members of one configuration share a grid, results of single launches
        state['variant'] = variant
        geom7 = conv.Geometry(1, Lx, (1, 7), (1, 1), (1, 1), (0, 3), False)
        items = [dict(x=x, w=w, geom=geom, bias=bias, res=res),
                 dict(x=x, w=(torch.randn(7, Co, Ci) / (7 * Ci) ** 0.5).bfloat16(), geom=geom7, bias=bias, in_slope=0.1),
                 dict(x=x[:1], w=w, geom=geom, res=res[:1])]
        conv._PLANS.clear()
        singles = [conv.conv_forward(**it) for it in items]
        n0 = lib.get().msmc_conv_launch_count()
        grouped = conv.conv_forward_group(items)
        assert lib.get().msmc_conv_launch_count() - n0 == 1
        assert b'conv_gather5_group_kernel' in lib.get().msmc_conv_last_kernel()
        for a, b in zip(grouped, singles):
            assert torch.equal(a, b)
        # outside the scope: a 1-tap layer -> MSMC_E_SHAPE surfaces as an error
        conv._PLANS.clear()
        with pytest.raises(RuntimeError, match='msmc_conv_gather'):
            _convcases.check_conv_case(('g5 k1', 1, 64, 128, 1, 40, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0),
                                       torch.bfloat16, 2e-2, 'cpu', parts=('fwd',))
    finally:
        conv._build_desc = real
        conv._PLANS.clear()


@pytest.mark.parametrize('variant', [56, 59, 60, 61, 63])
def test_gather_seventh_generation_variants(variant):
    """variants 56, 59, 60, 61, 63 (gather7.inc: eight waves of 64 x 64 -- 2 x 2 accumulators per wave --, the k16-steps of a stage split
    over 1 / 2 / 4 wave groups whose partial blocks meet in LDS by recursive halving, double-buffered super-stages of 1 or 2
    stages per barrier): every tile shape where it applies, forward and data gradient -- ragged pixel and channel tiles,
    dilation, stride, reflection, 2-D taps, two taps, odd and even stage counts against the super-stage length, several
    chunks with an input activation (in-place pass on the chunk whose tap 0 sits in the MIDDLE of a super-stage), more pixel
    tiles than one XCD group -- every epilogue operand against the second generation, grouped calls equal to single launches,
    1-tap layers refused"""
    from msmctts_amd.hip import conv, lib
    cases = [('g7 k3 64->128', 2, 64, 128, 1, 150, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
             ('g7 k11 d5 128->136', 1, 128, 136, 1, 70, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1),
             ('g7 k5x1 s3 64->256', 1, 64, 256, 40, 3, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
             ('g7 3x3 reflect s2 64->72', 1, 64, 72, 13, 18, (3, 3), (2, 2), (1, 1), (1, 1), True, 1.0),
             ('g7 k2 192->128', 1, 192, 128, 1, 45, (1, 2), (1, 1), (1, 1), (0, 1), False, 1.0),
             ('g7 k3 many tiles 64->128', 1, 64, 128, 1, 1300, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
             ('g7 ffn 256->512 relu', 2, 256, 512, 1, 100, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.0),
             ('g7 k3 192->64 lrelu chunk starts mid super-stage', 1, 192, 64, 1, 90, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
             ('g7 k5 128->128 lrelu', 2, 128, 128, 1, 75, (1, 5), (1, 1), (1, 2), (0, 4), False, 0.2),
             ('g7 k7 d3 64->64 thin', 1, 64, 64, 1, 700, (1, 7), (1, 1), (1, 3), (0, 9), False, 0.1),
             ('g7 k5x1 512->128 deep', 1, 512, 128, 30, 5, (5, 1), (1, 1), (1, 1), (2, 0), False, 0.2),
             ('g7 k3 88->136: the last chunk ends inside its 64 channels', 2, 88, 136, 1, 100, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
             ('g7 k3 200->72 relu, three whole chunks and eight channels', 1, 200, 72, 1, 90, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.0)]
    real = conv._build_desc
    state = {'variant': variant}

    def forced(*a, **k):
        d = real(*a, **k)
        if d.dtype == 1:
            d.variant = state['variant']
        return d
    conv._build_desc = forced
    ran = 0
    try:
        for case in cases:
            for part in ('fwd', 'dgrad'):
                conv._PLANS.clear()
                try:
                    _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', parts=(part,))
                    assert b'conv_gather7_kernel' in lib.get().msmc_conv_last_kernel(), (case[0], part)
                    ran += 1
                except RuntimeError as e:             # MSMC_E_SHAPE: this configuration does not apply to the layer
                    assert 'msmc_conv_gather' in str(e), e
        assert ran >= 3, (variant, ran)
        torch.manual_seed(0)
        B, Ci, Co, Lx = 2, 128, 264, 150
        geom = conv.Geometry(1, Lx, (1, 3), (1, 1), (1, 2), (0, 2), False)
        x = torch.randn(B, 1, Lx, Ci).bfloat16()
        w = (torch.randn(3, Co, Ci) / (3 * Ci) ** 0.5).bfloat16()
        bias, res, res2 = torch.randn(Co), torch.randn(B, 1, Lx, Co).bfloat16(), torch.randn(B, 1, Lx, Co).bfloat16()
        outs = []
        for v in (variant, 2):
            state['variant'] = v
            conv._PLANS.clear()
            fresh = conv.Geometry(1, Lx, (1, 3), (1, 1), (1, 2), (0, 2), False)         # (a geometry keeps its descriptors)
            outs.append(conv.conv_forward(x, w, fresh, bias=bias, in_slope=0.1, res=res, res2=res2, out_div=3.0, out_slope=0.2))
        assert b'conv_gather2' in lib.get().msmc_conv_last_kernel()
        assert _convcases.rel(outs[0], outs[1]) < 1e-2
        # grouped call: members of one configuration share a grid, results of single launches
        state['variant'] = variant
        geom3b = conv.Geometry(1, Lx, (1, 3), (1, 1), (1, 1), (0, 1), False)
        items = [dict(x=x, w=w, geom=geom, bias=bias, res=res),
                 dict(x=x, w=(torch.randn(3, Co, Ci) / (3 * Ci) ** 0.5).bfloat16(), geom=geom3b, bias=bias, in_slope=0.1),
                 dict(x=x[:1], w=w, geom=geom, res=res[:1])]
        conv._PLANS.clear()
        singles = [conv.conv_forward(**it) for it in items]
        n0 = lib.get().msmc_conv_launch_count()
        grouped = conv.conv_forward_group(items)
        assert lib.get().msmc_conv_launch_count() - n0 == 1
        assert b'conv_gather7_group_kernel' in lib.get().msmc_conv_last_kernel()
        for a, b in zip(grouped, singles):
            assert torch.equal(a, b)
        # outside the scope: a 1-tap layer -> MSMC_E_SHAPE surfaces as an error
        conv._PLANS.clear()
        with pytest.raises(RuntimeError, match='msmc_conv_gather'):
            _convcases.check_conv_case(('g7 k1', 1, 64, 128, 1, 40, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0),
                                       torch.bfloat16, 2e-2, 'cpu', parts=('fwd',))
    finally:
        conv._build_desc = real
        conv._PLANS.clear()


@pytest.mark.parametrize('g1v', [34, 35])
def test_gather_one_tap_gemm_variant_fp32(g1v):
    """variants 34 / 35 on fp32 operands (exact fp32 multiplies on v_mfma_f32_32x32x2_f32, fp32 output): the framed-DFT and
    mel-basis GEMM shapes of the spectral front-ends -- channel counts that are multiples of four only, ragged pixel and
    channel tiles, more than one round of four chunks -- forward and data gradient against PyTorch, every epilogue operand
    against the second generation"""
    from msmctts_amd.hip import conv, lib
    cases = [('g1f dft 204->200', 2, 204, 200, 1, 70, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0),
             ('g1f dft 480->484 ragged', 1, 480, 484, 1, 45, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0),
             ('g1f mel 516->80', 2, 516, 80, 1, 140, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0),
             ('g1f lrelu 64->72', 1, 64, 72, 1, 300, (1, 1), (1, 1), (1, 1), (0, 0), False, 0.1),
             ('g1f wide basis 512->1100 (channel tiles share an XCD)', 1, 512, 1100, 1, 130, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0)]
    real = conv._build_desc
    state = {'variant': g1v}

    def forced(*a, **k):
        d = real(*a, **k)
        d.variant = state['variant']
        return d
    conv._build_desc = forced
    try:
        for case in cases:
            for part in ('fwd', 'dgrad'):
                conv._PLANS.clear()
                _convcases.check_conv_case(case, torch.float32, 1e-5, 'cpu', parts=(part,))
                assert b'conv_gemm1_kernel<float' in lib.get().msmc_conv_last_kernel(), (case[0], part)
        torch.manual_seed(0)
        B, Ci, Co, Lx = 2, 132, 96, 150
        x = torch.randn(B, 1, Lx, Ci)
        w = torch.randn(1, Co, Ci) / Ci ** 0.5
        bias, res, res2 = torch.randn(Co), torch.randn(B, 1, Lx, Co), torch.randn(B, 1, Lx, Co)
        outs = []
        for v in (g1v, 2):
            state['variant'] = v
            conv._PLANS.clear()
            geom = conv.Geometry(1, Lx, (1, 1), (1, 1), (1, 1), (0, 0), False)
            outs.append(conv.conv_forward(x, w, geom, bias=bias, in_slope=0.1, res=res, res2=res2, out_div=3.0, out_slope=0.2))
        assert _convcases.rel(outs[0], outs[1]) < 1e-5
    finally:
        conv._build_desc = real
        conv._PLANS.clear()


def test_wgrad_direct_thin_layer_kernel():
    """variant 8 of the bf16 weight gradient (wgrad6.inc: one lattice point per work-item, a block of taps x output x input
    channels of dW in registers, DPP wave sums, one atomic per element and workgroup into the privatised copy): the thin
    discriminator layers -- 2->4 and 4->8 3x3 (stride 1 and 2, reflection), 1->16 5x1 stride 3, 8->16, a 1-channel output
    layer with 128 inputs, odd channel counts (blocks with dead lanes), several pixel splits -- weight and bias gradient
    against PyTorch; privatised copies summed; layers with too many blocks refused"""
    from msmctts_amd.hip import conv, lib
    cases = [('w6 mrd 2->4', 2, 2, 4, 21, 40, (3, 3), (1, 1), (1, 1), (1, 1), True, 1.0),
             ('w6 mrd 4->8 s2', 2, 4, 8, 21, 40, (3, 3), (2, 2), (1, 1), (1, 1), True, 0.2),
             ('w6 mrd 8->16', 1, 8, 16, 9, 50, (3, 3), (1, 1), (1, 1), (1, 1), True, 0.2),
             ('w6 mpd 1->16 s3', 2, 1, 16, 300, 3, (5, 1), (3, 1), (1, 1), (2, 0), False, 1.0),
             ('w6 post 128->1', 2, 128, 1, 12, 5, (3, 1), (1, 1), (1, 1), (1, 0), False, 0.2),
             ('w6 odd 3->5 k3', 1, 3, 5, 1, 700, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
             ('w6 many points 2->4', 1, 2, 4, 1, 5000, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0)]
    real = conv._build_desc
    state = {'variant': 8}

    def forced(*a, **k):
        d = real(*a, **k)
        if d.dtype == 1:
            d.variant = state['variant']
        return d
    conv._build_desc = forced
    try:
        for case in cases:
            conv._PLANS.clear()
            _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', parts=('wgrad',))
            assert b'conv_wgrad6_kernel' in lib.get().msmc_conv_last_kernel(), case[0]
        # privatised copies: the sum over the copies is the gradient
        torch.manual_seed(0)
        B, H, W, Ci, Co = 2, 11, 300, 2, 4
        geom = conv.Geometry(H, W, (3, 3), (1, 1), (1, 1), (1, 1), False)
        x, g = torch.randn(B, H, W, Ci).bfloat16(), torch.randn(B, H, W, Co).bfloat16()
        dw1, db1 = torch.zeros(9, Co, Ci), torch.zeros(Co)
        conv._PLANS.clear()
        conv.conv_wgrad(x, g, geom, 9, in_slope=1.0, dw=dw1, db=db1)
        dw8, db8 = torch.zeros(8, 9, Co, Ci), torch.zeros(8, Co)
        conv._PLANS.clear()
        conv.conv_wgrad(x, g, conv.Geometry(H, W, (3, 3), (1, 1), (1, 1), (1, 1), False), 9, in_slope=1.0, dw=dw8.view(-1),
                        db=db8.view(-1), copies=8)
        assert _convcases.rel(dw8.sum(0), dw1) < 1e-5 and _convcases.rel(db8.sum(0), db1) < 1e-5
        # too many blocks of dW: MSMC_E_SHAPE surfaces as an error
        conv._PLANS.clear()
        with pytest.raises(RuntimeError, match='wgrad'):
            _convcases.check_conv_case(('w6 big 128->128', 1, 128, 128, 1, 40, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
                                       torch.bfloat16, 2e-2, 'cpu', parts=('wgrad',))
    finally:
        conv._build_desc = real
        conv._PLANS.clear()


def test_gather_thin_channel_variant():
    """variant 50 (gather6.inc: Cin 8 / 16 / 32 -- the B fragment of the implicit GEMM is eight consecutive channels of one
    tap, loaded straight from global memory; weights in LDS; a wave walks tiles of 32 lattice points alone): the thin
    discriminator layers, forward and data gradient -- 3x3 stride 1 and 2 with reflection, 5x1 stride 3, a contraction that
    is not a multiple of sixteen (Cin 8 x 9 taps), two output blocks (Cout 64), Cout 8 (data gradient of an 8 -> 16
    layer), ragged last tile -- every epilogue operand against the second generation; grouped calls on one grid"""
    from msmctts_amd.hip import conv, lib
    cases = [('t6 mrd 8->16', 2, 8, 16, 21, 40, (3, 3), (1, 1), (1, 1), (1, 1), True, 1.0),
             ('t6 mrd 16->32 s2', 2, 16, 32, 21, 40, (3, 3), (2, 2), (1, 1), (1, 1), True, 0.2),
             ('t6 mrd 32->64', 1, 32, 64, 9, 50, (3, 3), (1, 1), (1, 1), (1, 1), True, 0.2),
             ('t6 mpd 16->64 s3', 2, 16, 64, 100, 3, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
             ('t6 k7 32->8', 1, 32, 8, 1, 333, (1, 7), (1, 1), (1, 1), (0, 3), False, 0.1),
             ('t6 k3 8->8 ragged', 1, 8, 8, 1, 77, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0),
             ('t6 mrd 4->8 s2', 2, 4, 8, 21, 40, (3, 3), (2, 2), (1, 1), (1, 1), True, 0.2),
             ('t6 mrd 2->4', 2, 2, 4, 13, 50, (3, 3), (1, 1), (1, 1), (1, 1), True, 1.0),
             ('t6 k5x1 s3 4->20', 1, 4, 20, 60, 3, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
             ('t6 k11 d5 64->64', 1, 64, 64, 1, 150, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1)]
    real = conv._build_desc
    state = {'variant': 50}

    def forced(*a, **k):
        d = real(*a, **k)
        if d.dtype == 1:
            d.variant = state['variant']
        return d
    conv._build_desc = forced
    ran = 0
    try:
        for case in cases:
            for part in ('fwd', 'dgrad'):
                conv._PLANS.clear()
                try:
                    _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', parts=(part,))
                    assert b'conv_gather6_kernel' in lib.get().msmc_conv_last_kernel(), (case[0], part)
                    ran += 1
                except RuntimeError as e:             # MSMC_E_SHAPE: outside the scope (e.g. the 64-channel data gradient)
                    assert 'msmc_conv_gather' in str(e), e
        assert ran >= 16, ran
        torch.manual_seed(0)
        B, H, W, Ci, Co = 2, 11, 30, 16, 40
        x = torch.randn(B, H, W, Ci).bfloat16()
        w = (torch.randn(9, Co, Ci) / (9 * Ci) ** 0.5).bfloat16()
        bias, res, res2 = torch.randn(Co), torch.randn(B, H, W, Co).bfloat16(), torch.randn(B, H, W, Co).bfloat16()
        outs = []
        for v in (50, 2):
            state['variant'] = v
            conv._PLANS.clear()
            geom = conv.Geometry(H, W, (3, 3), (1, 1), (1, 1), (1, 1), False)
            outs.append(conv.conv_forward(x, w, geom, bias=bias, in_slope=0.1, res=res, res2=res2, out_div=3.0, out_slope=0.2))
        assert _convcases.rel(outs[0], outs[1]) < 1e-2
        state['variant'] = 50
        items = []
        for Wd, ci in ((30, 16), (17, 4), (30, 32)):            # (members of different channel counts share the grid)
            geom = conv.Geometry(H, Wd, (3, 3), (1, 1), (1, 1), (1, 1), False)
            items.append(dict(x=torch.randn(B, H, Wd, ci).bfloat16(), w=(torch.randn(9, Co, ci) / (9 * ci) ** 0.5).bfloat16(),
                              geom=geom, bias=bias, out_slope=0.2))
        conv._PLANS.clear()
        singles = [conv.conv_forward(**it) for it in items]
        n0 = lib.get().msmc_conv_launch_count()
        grouped = conv.conv_forward_group(items)
        assert lib.get().msmc_conv_launch_count() - n0 == 1
        assert b'conv_gather6_group_kernel' in lib.get().msmc_conv_last_kernel()
        for a, b in zip(grouped, singles):
            assert torch.equal(a, b)
    finally:
        conv._build_desc = real
        conv._PLANS.clear()


@pytest.mark.parametrize('batch', [1, 2, 8])
def test_weight_gradients_waiting_in_the_bank_equal_immediate_ones(batch):
    """hip/convnet.py ConvBank.queue_wgrad: single-layer weight gradients go out as grouped calls of up to
    MSMC_WGRAD_BATCH members; a layer applied TWICE in one pass (the second application is queued while the first still
    waits) must flush first.  Compared with the stock operator chain and across batch sizes."""
    import torch.nn.functional as F
    from msmctts_amd.hip import convnet
    from msmctts_amd.networks.hifigan.common import ResBlock1
    keep = convnet.WGRAD_BATCH
    convnet.WGRAD_BATCH = batch
    try:
        torch.manual_seed(11)
        rb = ResBlock1(16, 3, (1, 3, 5))
        x = torch.randn(2, 16, 53, requires_grad=True)
        y = rb(rb(x))                                   # every layer twice
        go = torch.randn_like(y)
        (y * go).sum().backward()
        got = {n: p.grad.clone() for n, p in rb.named_parameters()}
        gx = x.grad.clone()
        rb.zero_grad()
        xr = x.detach().clone().requires_grad_(True)
        h = xr
        for _ in range(2):
            for c1, c2 in zip(rb.convs1, rb.convs2):
                t_ = F.conv1d(F.leaky_relu(h, 0.1), c1.weight(), c1.bias, 1, c1.padding, c1.dilation)
                h = F.conv1d(F.leaky_relu(t_, 0.1), c2.weight(), c2.bias, 1, c2.padding, c2.dilation) + h
        (h * go).sum().backward()
        _parity.close(y, h, 2e-4, what='out')
        _parity.close(gx, xr.grad, 2e-4, 1e-3, what='gx')
        for n, p in rb.named_parameters():
            _parity.close(got[n], p.grad, 2e-4, 2e-3, what=n)
        assert not any(items for _, items in rb._bank._pending_w.values()) if hasattr(rb, '_bank') else True
    finally:
        convnet.WGRAD_BATCH = keep




def test_triple_loss_kernel_matches_the_operator_chain():
    """csrc/losses.hip triple_loss_kernel on the interpreter (tests/_parity.py check_triple_loss)"""
    _parity.check_triple_loss('cpu')


def test_fft_stack_prologue_equals_the_operator_chain():
    _parity.check_fft_prologue('cpu')


def test_fft_stack_with_the_fused_prologue_equals_the_stack_called_with_positions():
    """FFTBlocks in bf16 on the HIP attention path: forward(seq, None, lengths=...) with MSMC_FFT_PROLOGUE on against
    forward(seq, positions) -- same bits out, same gradient in"""
    from msmctts_amd.networks.acoustic_models import transformer
    torch.manual_seed(9)
    stack = transformer.FFTBlocks(max_seq_len=100, n_layers=2, n_head=2, d_k=64, d_v=64, d_model=128, d_inner=256,
                                  fft_conv1d_kernel=3, fft_conv1d_padding=1, dropout=0.0, name='t', attn_dropout=0.0)
    stack.hip_dtype = torch.bfloat16
    stack.train()
    lengths = torch.tensor([37, 12, 1], dtype=torch.int32)
    T = 37
    steps = torch.arange(1, T + 1).unsqueeze(0)
    pos = steps * (steps <= lengths.unsqueeze(1))
    seq = torch.randn(3, T, 128)
    go = torch.randn(3, T, 128).to(torch.bfloat16)
    calls = []
    real = transformer.hipnorm.fft_prologue
    outs = []
    keep_flag = transformer.FFT_PROLOGUE
    try:
        for fused in (False, True):
            transformer.FFT_PROLOGUE = fused
            transformer.hipnorm.fft_prologue = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
            x = seq.clone().requires_grad_(True)
            out, keep = stack(x, None if fused else pos, lengths=lengths)
            (out * go).sum().backward()
            outs.append((out.detach(), keep, x.grad.clone()))
            stack.zero_grad()
    finally:
        transformer.FFT_PROLOGUE = keep_flag
        transformer.hipnorm.fft_prologue = real
    assert calls == [1]                                   # only the fused run took the one-launch path
    assert outs[0][0].dtype == torch.bfloat16 and torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


def test_layernorm_parameter_gradients_delivered_once_per_pass():
    """hip/norm.py: the dgamma / dbeta reductions of a backward pass leave in one msmc_add_ln_param_multi launch from an
    end-of-pass callback.  A LayerNorm applied TWICE in one pass (its second reduction goes out in a later launch of the same
    flush), a second backward without zero_grad (live gradients are added to, in the kernel), and MSMC_LN_PARAM_DEFER off --
    all against torch.nn.functional.layer_norm."""
    from msmctts_amd.hip import lib, norm
    torch.manual_seed(3)
    N, C = 45, 96
    x1, x2 = torch.randn(N, C), torch.randn(N, C)
    g1, g2 = torch.randn(N, C), torch.randn(N, C)

    def reference(passes):
        gm, bt = torch.full((C,), 1.25, requires_grad=True), torch.full((C,), -0.5, requires_grad=True)
        om, ob = torch.ones(C, requires_grad=True), torch.zeros(C, requires_grad=True)
        for _ in range(passes):
            a = torch.nn.functional.layer_norm(x1, (C,), gm, bt)
            b = torch.nn.functional.layer_norm(a + x2, (C,), gm, bt)          # the same parameters a second time
            c = torch.nn.functional.layer_norm(b, (C,), om, ob)
            ((a * g1).sum() + (c * g2).sum()).backward()
        return [t.grad.clone() for t in (gm, bt, om, ob)]

    def product(passes, defer):
        keep = norm.LN_PARAM_DEFER
        norm.LN_PARAM_DEFER = defer
        calls = []
        real = lib.get().msmc_add_ln_param_multi
        try:
            gm, bt = torch.full((C,), 1.25, requires_grad=True), torch.full((C,), -0.5, requires_grad=True)
            om, ob = torch.ones(C, requires_grad=True), torch.zeros(C, requires_grad=True)
            setattr(lib.get(), 'msmc_add_ln_param_multi', lambda items, n, st: (calls.append(n), real(items, n, st))[1])
            for _ in range(passes):
                a = norm.add_layer_norm(x1, None, gm, bt)
                b = norm.add_layer_norm(a, x2, gm, bt)
                c = norm.add_layer_norm(b, None, om, ob)
                ((a * g1).sum() + (c * g2).sum()).backward()
            return [t.grad.clone() for t in (gm, bt, om, ob)], calls
        finally:
            setattr(lib.get(), 'msmc_add_ln_param_multi', real)
            norm.LN_PARAM_DEFER = keep

    for passes in (1, 2):
        want = reference(passes)
        got, calls = product(passes, True)
        assert calls == [2, 1] * passes, calls        # per pass: {first use of (gm, bt), (om, ob)} then the second use of (gm, bt)
        for a, b in zip(got, want):
            assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max())), (passes, float((a - b).abs().max()))
        got0, calls0 = product(passes, False)
        assert calls0 == []
        for a, b in zip(got0, want):
            assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max()))


def test_layernorm_parameter_gradients_under_restricted_passes():
    """hip/norm.py with the deferred parameter-gradient launch on: ``backward(inputs=[...])`` delivers exactly the requested
    parameter gradients (none when only the activation is asked for), and ``torch.autograd.grad`` on the LayerNorm parameters
    gets their gradients (autograd's own route: that pass accumulates nothing into ``.grad``)."""
    from msmctts_amd.hip import norm
    torch.manual_seed(5)
    N, C = 37, 64
    x, g = torch.randn(N, C, requires_grad=True), torch.randn(N, C)
    gm, bt = torch.full((C,), 0.75, requires_grad=True), torch.full((C,), 0.25, requires_grad=True)
    ref = torch.nn.functional.layer_norm(x, (C,), gm, bt)
    want_gm, want_bt, want_x = torch.autograd.grad((ref * g).sum(), [gm, bt, x])
    assert norm.LN_PARAM_DEFER
    # both parameters requested (and nothing else): delivered by the end-of-pass launch
    (norm.add_layer_norm(x, None, gm, bt) * g).sum().backward(inputs=[gm, bt])
    _parity.close(gm.grad, want_gm, 2e-4, 1e-3, 'dgamma')
    _parity.close(bt.grad, want_bt, 2e-4, 1e-3, 'dbeta')
    assert x.grad is None
    gm.grad = bt.grad = None
    # one of the two: autograd's own route for it, nothing for the other
    (norm.add_layer_norm(x, None, gm, bt) * g).sum().backward(inputs=[gm])
    _parity.close(gm.grad, want_gm, 2e-4, 1e-3, 'dgamma alone')
    assert bt.grad is None and x.grad is None
    gm.grad = None
    # only the activation: no parameter gradient is computed or delivered
    (norm.add_layer_norm(x, None, gm, bt) * g).sum().backward(inputs=[x])
    _parity.close(x.grad, want_x, 2e-4, 1e-3, 'dx')
    assert gm.grad is None and bt.grad is None
    x.grad = None
    # torch.autograd.grad: the gradients come back on the edges, .grad stays untouched
    a, b = torch.autograd.grad((norm.add_layer_norm(x, None, gm, bt) * g).sum(), [gm, bt])
    _parity.close(a, want_gm, 2e-4, 1e-3, 'dgamma (autograd.grad)')
    _parity.close(b, want_bt, 2e-4, 1e-3, 'dbeta (autograd.grad)')
    (c,) = torch.autograd.grad((norm.add_layer_norm(x, None, gm, bt) * g).sum(), [bt])
    _parity.close(c, want_bt, 2e-4, 1e-3, 'dbeta alone (autograd.grad)')
    assert gm.grad is None and bt.grad is None and x.grad is None


def test_train_steps_match_reference_with_the_fused_fft_prologue():
    """the same reference fixture with MSMC_FFT_PROLOGUE on (FFT stacks called with lengths instead of positions)"""
    from msmctts_amd.networks.acoustic_models import transformer
    keep = transformer.FFT_PROLOGUE
    transformer.FFT_PROLOGUE = True
    try:
        _parity.check_train_steps('cpu')
    finally:
        transformer.FFT_PROLOGUE = keep


def test_bench_attributes_work_to_every_kernel_family_of_a_step():
    """bench.py's per-kernel table: every C-ABI helper call of a GAN-phase step (normalisation, attention is bf16-only and
    absent here, losses, spectral glue, reflect folds, VQ statistics, weight-norm passes, fused optimizer, deferred
    weight-gradient second stage) is wrapped with a work model whose signature matches the call -- a mismatch would only
    surface as a TypeError on the GPU box -- and every launch the library logs ends up with flops or bytes."""
    import bench
    from msmctts_amd.hip import lib
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    L = lib.get()
    saved = {name: getattr(L, name) for name in list(bench.abi_work_models()) +
             ['msmc_wn_prepare_multi_tiles', 'msmc_wn_backward_multi_rows', 'msmc_opt_clip_adamw']}
    timer = bench.KernelTimer()
    try:
        for name, work in bench.abi_work_models().items():
            timer.wrap_abi(L, name, work)
        seen = []
        for name in ('msmc_wn_prepare_multi_tiles', 'msmc_wn_backward_multi_rows', 'msmc_opt_clip_adamw'):
            timer.wrap_abi(L, name, (lambda nm: lambda *a: (seen.append(nm), (0.0, 64.0))[1])(name))
        cfg, task = _parity.build_small('cpu')
        tr = build_trainer(cfg, task, num_gpus=0, rank=0)
        tr.model = task
        tr.optimizer = build_optimizer(task, cfg.optimizer)
        z = _parity.load_npz('small_steps.npz')
        fw = [tuple(int(v) for v in r) for r in z['windows']]
        tr.random_select = lambda ml: (fw, [(s * 300, e * 300) for s, e in fw])
        batch = {k[len('batch.'):]: _parity.t(v) for k, v in z.items() if k.startswith('batch.')}
        task.zero_grad()
        timer.start(L)
        tr.train_step(batch, 6)
        timer.stop()
        summary = timer.summary()
    finally:
        for name, fn in saved.items():
            setattr(L, name, fn)
    assert set(seen) == {'msmc_wn_prepare_multi_tiles', 'msmc_wn_backward_multi_rows', 'msmc_opt_clip_adamw'}
    called = set(timer.shapes)
    for family in ('msmc_add_ln_fwd', 'msmc_add_ln_bwd', 'msmc_add_ln_param_multi', 'msmc_l1_multi_fwd_ws', 'msmc_mse_const_multi_bwd', 'msmc_spectral_multi',
                   'msmc_spec_mag_bwd', 'msmc_vq_backward', 'msmc_vq_prepare', 'msmc_tanh_fwd', 'msmc_gate_bwd'):
        assert family in called, (family, sorted(called))
    assert summary, 'the interpreter build logs its launches too'
    # convolution / search launches get their work from bench.py's host-level wrappers (not installed here); everything
    # else must carry work now
    bare = [k for k, v in summary.items() if v['flops'] == 0 and v['bytes'] == 0 and not k.startswith(('conv_', 'vq_search'))]
    assert not bare, bare


def test_masked_mean_and_non_atomic_colsum_match_stock_operators():
    _parity.check_masked_mean_and_colsum('cpu')


def test_multi_tensor_gan_loss_kernels_match_stock_operators():
    _parity.check_gan_loss_kernels('cpu')


def test_split_bf16_constant_matrix_gemm_and_spectral_chain():
    _parity.check_split_constant_gemm('cpu')


def test_wave_fan_out_matches_cast_pad_and_gradient_accumulation():
    _parity.check_wave_fan('cpu')


def test_clean_weight_banks_skip_their_refresh_and_notice_every_kind_of_update():
    """ConvBank.prepare launches nothing while the bank is clean (hip/convnet.py SKIP_CLEAN_PREPARE) and refreshes the
    kernel-layout weights after (a) a HipAdamW step -- a raw-pointer update no version counter sees --, (b) an in-place torch
    operation on a weight (load_state_dict, init), (c) the first use; the outputs follow the parameters in every case."""
    from msmctts_amd.hip import convnet, lib
    from msmctts_amd.networks.hifigan.common import ResBlock1
    from msmctts_amd.trainers.optimizers.hip_adamw import HipAdamW
    L = lib.get()
    torch.manual_seed(1)
    blk = ResBlock1(16, 3, (1, 3, 5))
    x = torch.randn(2, 16, 40)

    def prepares(fn):
        L.msmc_prof_enable(1)
        out = fn()
        names = []
        import ctypes
        buf, ms = ctypes.create_string_buffer(128), ctypes.c_float()
        for i in range(L.msmc_prof_count()):
            L.msmc_prof_read(i, buf, 128, ctypes.byref(ms))
            names.append(buf.value.decode())
        L.msmc_prof_enable(0)
        return out, sum(1 for n in names if n.startswith('wn_layout'))          # (one layout pass per refresh)

    y0, n0 = prepares(lambda: blk(x))
    assert n0 == 1                                           # first use: the images are built
    y1, n1 = prepares(lambda: blk(x))
    assert n1 == 0 and torch.equal(y0, y1)                   # clean: nothing to refresh
    opt = HipAdamW(list(blk.parameters()), lr=1e-2)
    blk(x).square().mean().backward()
    opt.step()                                               # raw-pointer update of weight_v / weight_g
    y2, n2 = prepares(lambda: blk(x))
    assert n2 == 1 and not torch.equal(y2, y1)
    with torch.no_grad():
        blk.convs1[0].weight_v.mul_(1.5)                     # in-place torch operation: the version counter moves
    y3, n3 = prepares(lambda: blk(x))
    assert n3 == 1 and not torch.equal(y3, y2)
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    with torch.no_grad():
        for p in blk.parameters():
            p.add_(0.01)
    blk.load_state_dict(sd)
    y4, n4 = prepares(lambda: blk(x))
    assert n4 == 1 and torch.allclose(y4, y3, atol=1e-6)
    keep = convnet.SKIP_CLEAN_PREPARE
    try:
        convnet.SKIP_CLEAN_PREPARE = False                   # the A/B switch: every forward refreshes
        _, n5 = prepares(lambda: blk(x))
        assert n5 == 1
    finally:
        convnet.SKIP_CLEAN_PREPARE = keep
