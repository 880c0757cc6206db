"""GPU (-m gpu): the product path over the real gfx950 library against
  (1) the golden fixtures generated from the reference (tests/golden),
  (2) the oracle on the same seeded inputs at sizes it finishes in seconds,
  (3) size-independent properties at BASELINE.json's full sizes.
Bars: VQ indices / quantised values bit-exact vs the C oracle; fp32 module outputs, losses and
gradients within 1e-3 of the reference; post-step VQ buffers within 1e-5."""
import numpy as np
import pytest
import torch

import _parity

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module', autouse=True)
def real_library():
    from msmctts_amd.hip import lib
    assert lib.backend() == 'gfx950', 'GPU tests must run on the real HIP library'
    torch.backends.cuda.matmul.allow_tf32 = False
    yield


def test_vq_fixture_cases():
    _parity.check_vq_cases(DEV)


def test_modules_match_reference():
    _parity.check_modules(DEV)


def test_train_steps_match_reference():
    _parity.check_train_steps(DEV)


def test_predictor_step_matches_reference():
    """BASELINE config #4 (predictor training against the frozen autoencoder)"""
    _parity.check_predictor_step(DEV)


def test_window_gather_and_output_tanh_match_the_operator_chains():
    _parity.check_window_gather_and_output_tanh(torch.device(DEV))


def test_two_autograd_graphs_over_one_bank_back_propagated_one_after_the_other():
    _parity.check_two_graphs_over_one_bank(DEV)


def test_sum_dropout_add_row_mask_kernels_and_shared_input_gradients():
    _parity.check_glue_kernels(DEV)


def test_direct_kernels_with_every_epilogue_operand_at_once():
    _parity.check_direct_kernels_with_every_epilogue_operand(DEV)


def test_output_projection_inside_the_add_layernorm_launch():
    _parity.check_fc_add_ln(DEV)


def test_front_end_chains_in_lock_step_equal_the_chains_one_by_one():
    _parity.check_fronts_lockstep(torch.device(DEV))


def test_weight_images_in_one_tiled_pass_match_the_definition():
    """msmc_wn_prepare_multi_tiles (round 6: both kernel layouts from one read of the parameters) on the GPU"""
    _parity.check_weight_image_tiles(torch.device(DEV))


@pytest.mark.parametrize('shortlist', [None, True, False], ids=['product', 'shortlist-kernel', 'exact-kernel'])
@pytest.mark.parametrize('H,K,D,N', [(1, 64, 256, 777), (4, 64, 256, 6400), (4, 256, 256, 1600), (8, 512, 256, 530),
                                     (4, 16, 32, 51), (2, 48, 24, 1), (4, 64, 256, 16), (4, 64, 256, 17)])
def test_vq_search_bit_exact_vs_c_oracle(H, K, D, N, shortlist):
    """``product``: the kernel the product picks for the shape and size; ``shortlist-kernel``: csrc/vq_shortlist.inc forced
    (d in {32, 64}; skipped elsewhere); ``exact-kernel``: the register-resident / LDS-tile exact kernels forced"""
    from msmctts_amd.hip import vq
    from oracle import cvq
    rng = np.random.default_rng(H * 1000 + K + N)
    x = rng.standard_normal((N, D)).astype(np.float32)
    e = rng.standard_normal((H, D // H, K)).astype(np.float32)
    want = cvq.search(x, e)
    et, en = vq.vq_prepare(torch.from_numpy(e).to(DEV))
    assert np.array_equal(et.cpu().numpy(), e.transpose(0, 2, 1))
    if shortlist and getattr(et, 'shortlist_image', None) is None:
        pytest.skip('the shortlist kernel does not take this shape')
    q, d, i = vq.vq_search(torch.from_numpy(x).to(DEV), et, en, shortlist=shortlist)
    assert np.array_equal(i.cpu().numpy(), want['ind'])
    assert np.array_equal(q.cpu().numpy(), want['quant'])
    assert np.array_equal(d.cpu().numpy(), want['diff'])


@pytest.mark.parametrize('H,K', [(4, 64), (4, 256), (8, 512), (2, 48)])
def test_vq_shortlist_is_bit_identical_to_the_exact_kernel(H, K):
    """csrc/vq_shortlist.inc against the exact kernel on 2^18 frames (every output, bit for bit): Gaussian frames,
    frames that ARE codewords, runs of two and three duplicate codewords, a near-duplicate at relative distance 1e-7,
    frames 1e-4 from a duplicated codeword; plus rows of it against the C oracle.  The planted ties must reach the exact
    paths; Gaussian data must mostly be decided by the shortlist."""
    from msmctts_amd.hip import lib, vq
    from oracle import cvq
    N, d = 1 << 18, 64 if H < 8 else 32
    D = H * d
    g = torch.Generator().manual_seed(K)
    e = torch.randn(H, d, K, generator=g)
    e[:, :, 7] = e[:, :, 3]
    e[:, :, 11] = e[:, :, 3]
    e[:, :, 9] = e[:, :, 5] * (1 + 1e-7)
    x = torch.randn(N, D, generator=g)
    x[:4096] = e[:, :, 3].reshape(1, D) + torch.randn(4096, D, generator=g) * 1e-4
    x[8192:8192 + K] = e.permute(2, 0, 1).reshape(K, D)                    # every codeword (of all heads at once) as a frame
    x[16384:16384 + 64] = e[:, :, 5].reshape(1, D)
    e, x = e.to(DEV), x.to(DEV)
    et, en = vq.vq_prepare(e)
    assert getattr(et, 'shortlist_image', None) is not None
    vq.SLOW_COUNT = torch.zeros(2, dtype=torch.int64, device=DEV)
    try:
        q, df, i = vq.vq_search(x, et, en, shortlist=True)
        assert lib.get().msmc_vq_last_kernel() == b'vq_search_sl_kernel'
        slow = vq.SLOW_COUNT.tolist()
        vq.SLOW_COUNT.zero_()
        e_clean = torch.randn(H, d, K, generator=g).to(DEV)                # (no planted duplicates)
        etc, enc = vq.vq_prepare(e_clean)
        qc, dfc, ic = vq.vq_search(x[32768:], etc, enc, shortlist=True)    # the Gaussian part alone
        slow_gauss = vq.SLOW_COUNT.tolist()
    finally:
        vq.SLOW_COUNT = None
    q0, df0, i0 = vq.vq_search(x, et, en, shortlist=False)
    assert lib.get().msmc_vq_last_kernel() != b'vq_search_sl_kernel'
    assert torch.equal(i, i0) and torch.equal(q, q0) and torch.equal(df, df0)
    qc0, dfc0, ic0 = vq.vq_search(x[32768:], etc, enc, shortlist=False)
    assert torch.equal(ic, ic0) and torch.equal(qc, qc0) and torch.equal(dfc, dfc0)
    assert (i[:4096] == 3).all() and not ((i == 7) | (i == 11)).any()
    own = torch.where(torch.isin(torch.arange(K), torch.tensor([7, 11])), 3, torch.arange(K)).to(DEV).unsqueeze(1)
    clear = ~torch.isin(torch.arange(K), torch.tensor([5, 9])).to(DEV)     # (5 / 9: a 1e-7 pair, decided by fp32 rounding)
    assert (i[8192:8192 + K][clear] == own[clear]).all()
    rows = torch.cat((torch.arange(0, 4096, 37), torch.arange(8192, 8192 + K), torch.arange(16384, 16448),
                      torch.arange(32768, N, 1031))).to(DEV)
    w = cvq.search(x[rows].cpu().numpy(), e.cpu().numpy())
    assert np.array_equal(i[rows].cpu().numpy(), w['ind']) and np.array_equal(q[rows].cpu().numpy(), w['quant'])
    assert np.array_equal(df[rows].cpu().numpy(), w['diff'])
    tiles = (N - 32768) // 16 * H
    assert slow[1] >= 4096 // 16 * H and slow[0] >= 1
    assert slow_gauss[0] <= 0.3 * tiles and slow_gauss[1] <= 0.01 * tiles, (slow_gauss, tiles)


def test_vq_search_near_ties_and_duplicates():
    from msmctts_amd.hip import vq
    from oracle import cvq
    rng = np.random.default_rng(0)
    H, K, d = 4, 64, 64
    e = rng.standard_normal((H, d, K)).astype(np.float32)
    e[:, :, 7] = e[:, :, 3]                                  # exact duplicate -> first-minimum rule picks 3
    e[:, :, 9] = e[:, :, 5] * np.float32(1 + 1e-7)
    x = rng.standard_normal((2048, H * d)).astype(np.float32)
    x[:512] = np.tile(e[:, :, 3].reshape(-1), (512, 1)) + rng.standard_normal((512, H * d)).astype(np.float32) * 1e-4
    want = cvq.search(x, e)
    et, en = vq.vq_prepare(torch.from_numpy(e).to(DEV))
    q, dd, i = vq.vq_search(torch.from_numpy(x).to(DEV), et, en)
    assert np.array_equal(i.cpu().numpy(), want['ind'])
    assert (want['ind'][:512] == 3).all() and not (want['ind'] == 7).any()


def test_vq_search_vs_torch_oracle_gap_contract():
    """Against the plain-PyTorch oracle (BLAS accumulation order): identical wherever the fp64 top-2 gap
    exceeds 1e-4 * scale (SURVEY.md section 7 'hard parts')."""
    from msmctts_amd.hip import vq
    from oracle.vq import np_search
    rng = np.random.default_rng(5)
    H, K, D, N = 4, 256, 256, 20000
    x = rng.standard_normal((N, D)).astype(np.float32)
    e = rng.standard_normal((H, D // H, K)).astype(np.float32)
    et, en = vq.vq_prepare(torch.from_numpy(e).to(DEV))
    _, _, i = vq.vq_search(torch.from_numpy(x).to(DEV), et, en)
    got = i.cpu().numpy()
    ref = np_search(x, [e[h] for h in range(H)])
    bad = np.argwhere(got != ref)
    for n, h in bad:
        xh = x[n, h * 64:(h + 1) * 64].astype(np.float64)
        dist = ((xh[:, None] - e[h].astype(np.float64)) ** 2).sum(0)
        top = np.sort(dist)[:2]
        assert top[1] - top[0] <= 1e-4 * max(1.0, top[0]), (n, h, top)
    assert len(bad) <= 5


def test_vq_full_size_properties():
    """BASELINE sizes (N = 2^20 frames, D=256): idempotence and index range (no oracle at this size)."""
    from msmctts_amd.hip import vq
    N, D, H, K = 1 << 20, 256, 4, 256
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, D, generator=g).to(DEV)
    e = torch.randn(H, D // H, K, generator=g).to(DEV)
    et, en = vq.vq_prepare(e)
    q, d, i = vq.vq_search(x, et, en)
    assert int(i.min()) >= 0 and int(i.max()) < K
    gathered = torch.stack([e[h].t()[i[:, h]] for h in range(H)], 1).reshape(N, D)
    assert (q - gathered).abs().max().item() <= 1e-5 * 8      # x + (e - x) is e up to one rounding
    q2, d2, i2 = vq.vq_search(gathered, et, en)                # quantising codewords returns them: idempotent
    assert torch.equal(i2, i)
    assert d2.max().item() <= 1e-9
    want = ((gathered - x) ** 2).reshape(N, H, D // H).mean(1)
    assert (d - want).abs().max().item() <= 1e-4
    # a sample of rows against the C oracle, bit for bit
    from oracle import cvq
    rows = torch.arange(0, N, 4099, device=DEV)
    w = cvq.search(x[rows].cpu().numpy(), e.cpu().numpy())
    assert np.array_equal(i[rows].cpu().numpy(), w['ind'])


def test_vq_ema_matches_torch_oracle_large():
    from msmctts_amd.hip import vq
    from oracle.vq import multi_head_quantize
    B, T, D, H, K = 16, 400, 256, 4, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, D, generator=g)
    ln = torch.randint(T // 2, T + 1, (B,), generator=g)
    e = torch.randn(H, D // H, K, generator=g)
    heads = [(e[h].clone(), torch.rand(K, generator=g) * 5, e[h].clone() * 1.5) for h in range(H)]
    cs0 = torch.stack([h[1] for h in heads]).clone()
    ea0 = torch.stack([h[2] for h in heads]).clone()
    q, dd, ind = multi_head_quantize(x, ln, heads, True)
    emb, cs, ea = e.clone().to(DEV), cs0.to(DEV), ea0.to(DEV)
    et, en = vq.vq_prepare(emb)
    _, _, gi = vq.vq_search(x.to(DEV), et, en)
    assert torch.equal(gi.cpu(), ind)
    vq.vq_ema_update(x.to(DEV), gi, ln.to(DEV), emb, cs, ea, 0.99, 1e-5)
    for h in range(H):
        _parity.close(cs[h], heads[h][1], 1e-5, 1e-6, 'cluster_size')
        _parity.close(ea[h], heads[h][2], 1e-5, 1e-5, 'embed_avg')
        _parity.close(emb[h], heads[h][0], 1e-5, 1e-5, 'embed')
    emb2, cs2, ea2 = e.clone().to(DEV), cs0.to(DEV), ea0.to(DEV)
    vq.vq_ema_update(x.to(DEV), gi, ln.to(DEV), emb2, cs2, ea2, 0.99, 1e-5)
    assert torch.equal(emb, emb2) and torch.equal(ea, ea2)            # deterministic reduction order


def test_no_cpu_fallback():
    from msmctts_amd.hip import vq
    with pytest.raises(RuntimeError):
        vq.vq_prepare(torch.randn(1, 4, 16))


def test_graphed_step_matches_eager():
    """hipGraph replay of the GAN-phase step (three captured segments) against the eager step: same
    weights, batch and windows -> same losses and parameters (the small model has no dropout)."""
    graphed_vs_eager()


def test_graphed_warmup_phase_matches_eager_and_hands_over_to_the_gan_phase():
    """the warm-up phase (iteration < warmup_steps: no vocoder, no discriminator -- the reference trains 50 000 steps in it)
    replayed from its own hipGraphs against the eager steps, then -- same trainer -- the switch through the one eager step at
    iteration == warmup_steps into the captured GAN phase: two sets of graphs, static gradient tensors and optimizer tables alive
    side by side"""
    graphed_vs_eager(iterations=(0, 1), warmup_steps=4)
    graphed_vs_eager(iterations=(2, 3, 4, 5, 6), warmup_steps=4)


def graphed_vs_eager(arm_reducer=False, exchange='serial', iterations=(6, 7), warmup_steps=None):
    import random
    from msmctts_amd.synthetic import make_batch
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    batch = make_batch(3, 24, 80, 300, seed=5, device=DEV)
    batch['mel_length_host'] = batch['mel_length'].tolist()
    results = []
    for graphed in (False, True):
        cfg, task = _parity.build_small(DEV)
        tr = build_trainer(cfg, task, num_gpus=0, rank=0)
        tr.model = task
        if arm_reducer:       # gradient exchange between the graph segments / from the eager hooks
            from msmctts_amd.distributed.distributed import apply_gradient_allreduce
            apply_gradient_allreduce(task)
        tr.optimizer = build_optimizer(task, cfg.optimizer, capturable=True)
        tr.use_graphs = graphed
        tr.graph_exchange = exchange
        tr.rng = random.Random(3)
        if warmup_steps is not None:
            tr.warmup_steps = warmup_steps
        logs = []
        for it in iterations:
            if not tr.replays(it):               # (a replayed step owns static gradient buffers; capture happens on first use,
                task.zero_grad()                 #  its eager warm-up is rolled back)
            log = tr.train_step(batch, it)
            logs.append({k: float(v) for k, v in log['loss'].items()})
        if graphed and warmup_steps is not None:
            # the GAN phase captured: the warm-up phase's graphs (pool, static batch, optimizer tables) were released first
            gan = max(iterations) > warmup_steps
            assert (tr._graphs is not None) == gan and (tr._graphs_warm is None) == gan
        log, log2 = logs[0], logs[-1]
        results.append((log, log2, {k: v.detach().clone() for k, v in task.state_dict().items()}))
    (e1, e2, es), (g1, g2, gs) = results
    assert set(e1) == set(g1)
    for a, b in ((e1, g1), (e2, g2)):
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-3 * max(1.0, abs(a[k])), (k, a[k], b[k])
    for k in es:
        if es[k].dtype.is_floating_point:
            _parity.close(gs[k], es[k], 2e-3, 1e-3, k)


def test_rccl_gradient_reducer_single_rank_matches_reference():
    """The data-parallel path on the real backend: RCCL process group of one rank, bucketed reducer armed
    (autograd hooks + the HIP conv banks' hand-delivered gradients, collectives on RCCL's stream next to the
    multi-stream backward).  Averaging over one rank is the identity, so the golden step must still match; then
    the hipGraph step (capture with the process group alive, flat all-reduce between the replayed segments)
    against the eager step; then the same with the OVERLAPPED graph-mode exchange (the reducer's bucketed all-reduces
    captured into the segments on RCCL's stream, joined at the segment's end).  The subprocess has a hard timeout: a
    collective that wedges inside a capture fails the test instead of hanging the suite."""
    import subprocess, sys, os, socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch, torch.distributed as dist\n"
        "sys.path[:0] = [%r, %r, %r]\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d', world_size=1, rank=0)\n"
        "import _parity\n"
        "_parity.check_train_steps('cuda:0', arm_reducer=True)\n"
        "import test_gpu_parity\n"
        "test_gpu_parity.graphed_vs_eager(arm_reducer=True)\n"
        "test_gpu_parity.graphed_vs_eager(arm_reducer=True, exchange='overlap')\n"
        "dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()\n"
        "print('REDUCER-OK')\n" % (root, os.path.join(root, 'msmc-tts_amd'), os.path.join(root, 'tests'), port))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and 'REDUCER-OK' in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_reducer_waits_for_the_stream_a_bank_delivered_its_gradients_on():
    """A bucket that mixes convolution-bank parameters (gradients delivered EARLY, from a side stream of the bank's own:
    hip/convnet.py FINISH_SIDE) with parameters of stock modules (ready on the calling stream) must not be concatenated
    before the side stream has written the bank's gradients (round-4 review).  ``_parity.check_reducer_stream_order``: a
    stock ``nn.Linear`` in front of a one-layer convolution bank, ONE bucket, the bank's early delivery held back 0.1 s on its
    stream, its gradient buffers poisoned with NaN beforehand -- the bucket completes on the calling stream when the Linear's
    gradients arrive.  RCCL at world size 1 (averaging is the identity) in a subprocess with a hard timeout.
    ``tools/probes/reducer_race_probe.py`` is the same scenario with round 4's reducer."""
    import subprocess, sys, os, socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch, torch.distributed as dist\n"
        "sys.path[:0] = [%r, %r, %r]\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d', world_size=1, rank=0)\n"
        "import _parity\n"
        "worst = _parity.check_reducer_stream_order('cuda:0')\n"
        "dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()\n"
        "print('WORST %%.3e' %% worst)\n"
        "assert worst < 1e-5, worst\n"
        "print('ORDERED-OK')\n" % (root, os.path.join(root, 'msmc-tts_amd'), os.path.join(root, 'tests'), port))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and 'ORDERED-OK' in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_multi_resolution_stft_loss_matches_reference():
    """SURVEY 8a L2 on the HIP spectral front-end (framing, windowed-DFT GEMM, magnitude) against the fixture."""
    _parity.check_mr_stft(DEV)


def test_vq_edge_cases_empty_single_frame_zero_length_ragged():
    _parity.check_vq_edge_cases(DEV)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_fused_add_layernorm_gate_tanh_match_torch(dtype, tol):
    """csrc/norm.hip on the device against the stock operator chains (PyTorch-ROCm is the checker here)"""
    _parity.check_norm_kernels(DEV, dtype, tol)


def test_fft_stack_prologue_is_bit_identical_to_the_operator_chain():
    """csrc/norm.hip fft_prologue_kernel (MSMC_FFT_PROLOGUE=1, off by default) on the device, small and at the bench
    configuration's size"""
    _parity.check_fft_prologue(DEV)
    _parity.check_fft_prologue(DEV, B=16, T=400, C=256)


def test_predictor_step_replayed_from_graphs_matches_eager():
    """BASELINE config #4 on the graph path (round 5): PredictorTrainer.use_graphs -- the frozen analysis, the predictor, its
    losses and the backward pass as one hipGraph, clipping + Adam as a second -- against the eager trainer, three steps"""
    _parity.check_predictor_graphed_vs_eager(DEV)


def test_triple_loss_kernel_matches_the_operator_chain():
    """csrc/losses.hip triple_loss_kernel on the device against the stock operator chain (values and gradients)"""
    _parity.check_triple_loss(DEV)


def test_hip_adamw_matches_torch_adamw_with_clipping():
    _parity.check_hip_adamw(DEV)


def test_emb_autoencoder_matches_reference():
    """the QS-TTS synthesiser MSMCVQGANEmb against the reference's own module"""
    _parity.check_emb_autoencoder(DEV)


def test_hip_adamw_resume_then_capture_rollback_keeps_loaded_moments():
    _parity.check_hip_adamw_resume_then_capture_rollback(DEV)


def test_predictor_steps_draw_fresh_dropout_masks():
    _parity.check_predictor_dropout_masks_advance(DEV)


def test_codebook_statistics_split_update_is_the_fused_update():
    _parity.check_codebook_split_update(DEV)


def test_resblock_standalone_matches_stock_operators():
    _parity.check_resblock_standalone(DEV)


def test_inference_glue_matches_reference():
    _parity.check_inference(DEV)


def test_attention_kernels_match_the_reference_chain():
    _parity.check_attention(DEV)


def test_wave_exchange_primitives():
    """the shortlist search merges its per-lane candidates over lane groups with the gfx950 row / half swaps
    (v_permlane16_swap / v_permlane32_swap behind wave_xor16 / wave_xor32): a search whose winner, runner-up and third
    sit in DIFFERENT lane groups of every frame must still come out right -- codeword k belongs to lane group (k >> 2) & 3"""
    from msmctts_amd.hip import vq
    from oracle import cvq
    H, K, d, N = 4, 64, 64, 4096
    rng = np.random.default_rng(7)
    e = (rng.standard_normal((H, d, K)) * 4).astype(np.float32)
    x = np.empty((N, H * d), np.float32)
    picks = rng.integers(0, K, size=(N, H, 3))
    for h in range(H):                               # frame = mix of three codewords from (mostly) different lane groups
        a, b, c = (e[h][:, picks[:, h, j]].T for j in range(3))
        x[:, h * d:(h + 1) * d] = 0.5 * a + 0.3 * b + 0.2 * c + rng.standard_normal((N, d)).astype(np.float32) * 0.05
    want = cvq.search(x, e)
    et, en = vq.vq_prepare(torch.from_numpy(e).to(DEV))
    q, df, i = vq.vq_search(torch.from_numpy(x).to(DEV), et, en, shortlist=True)
    assert np.array_equal(i.cpu().numpy(), want['ind']) and np.array_equal(q.cpu().numpy(), want['quant'])
    assert len(np.unique((want['ind'] >> 2) & 3)) == 4


def test_masked_mean_and_non_atomic_colsum_match_stock_operators():
    _parity.check_masked_mean_and_colsum(DEV)


def test_multi_tensor_gan_loss_kernels_match_stock_operators():
    _parity.check_gan_loss_kernels(DEV)


def test_split_bf16_constant_matrix_gemm_and_spectral_chain():
    _parity.check_split_constant_gemm(DEV)


def test_wave_fan_out_matches_cast_pad_and_gradient_accumulation():
    _parity.check_wave_fan(DEV)


def test_bench_two_ranks_sharing_the_gpu_run_the_whole_data_parallel_path():
    """``python bench.py --gpus 2`` the way the driver starts the one-GPU line (WORLD_SIZE unset: the launcher starts the
    ranks), on a ONE-GPU box: ``--share-gpu --backend gloo`` puts both ranks on the device and exchanges over gloo, so every
    N > 1 branch of bench.py and of the trainer runs -- start-up broadcast, capture with a process group alive, the flat
    gradient exchange between the replayed segments, barrier + max-over-ranks timing, one rank-0 line.  The numbers are not a
    measurement (the ranks share the chip); the line's shape and the ranks' lock step are what is checked."""
    import json, subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--share-gpu', '--backend', 'gloo', '--batch', '4',
                          '--steps', '3', '--warmup', '2', '--no-microbench', '--cpu-steps', '0', '--fp32-steps', '0',
                          '--kernel-timing-steps', '0', '--warmup-phase-steps', '0', '--stall-timeout', '240', '--job-timeout', '500'],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['world_size_seen'] == 2 and 'gloo' in line['backend']
    assert len(line['per_rank_ms_per_step']) == 2 and all(t > 0 for t in line['per_rank_ms_per_step'])
    assert line['config']['global_batch'] == 8 and line['config']['gradient_exchange'] == 'serial' and line['config']['parallelism'] == 'dp2'
    assert line['value'] > 0 and all(v == v for v in line['losses'].values())


def test_bench_two_ranks_time_the_secondary_exchange_modes_under_the_guard():
    """``--exchange all`` (what ``auto`` resolves to at N > 1 over RCCL) on the shared GPU over gloo: the headline's reductions
    over ranks happen BEFORE the secondary modes, the bf16-wire mode is timed after it under ``Watchdog.guard``, the third mode
    as a child job of rank 0, all three reported in ``exchange_modes``; one line, exit code 0."""
    import json, subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    # the third mode runs as a CHILD job of rank 0 after the other rank has left (over RCCL: the overlapped exchange; here, on one
    # GPU, the test hook makes the child a gloo / serial job -- what is checked is the orchestration and the merged line)
    env['MSMC_BENCH_CHILD_ARGS'] = '--share-gpu --backend gloo --exchange serial'
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--share-gpu', '--backend', 'gloo', '--batch', '4',
                          '--steps', '3', '--warmup', '2', '--no-microbench', '--cpu-steps', '0', '--fp32-steps', '0', '--exchange', 'all',
                          '--kernel-timing-steps', '0', '--warmup-phase-steps', '0', '--stall-timeout', '240', '--job-timeout', '500'],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    modes = line['exchange_modes']
    assert set(modes) == {'serial', 'serial_bf16_wire', 'overlap'} and all(m.get('ms_per_step', 0) > 0 for m in modes.values()), modes
    assert 'a job of its own' in modes['overlap']['note'] and len(modes['overlap']['per_rank_ms_per_step']) == 2
    assert abs(line['ms_per_step'] - modes['serial']['ms_per_step']) < 1e-6 and line['config']['gradient_exchange'] == 'serial'
    assert line['n_gpus'] == 2 and line['value'] > 0 and all(v == v for v in line['losses'].values())
