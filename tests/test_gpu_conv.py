"""GPU (-m gpu): the implicit-GEMM convolution kernels (csrc/conv.hip) at the CSMSC layer shapes against
PyTorch-ROCm's own convolutions on the same device (fp32 reference of the same op), forward, data gradient
and weight gradient; fp32 kernels within 2e-4 of the output scale, bf16 kernels within 2e-2."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def cl(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2)


def rel(a, b):
    return (a.float() - b.float()).abs().max().item() / max(1e-6, b.float().abs().max().item())


# (name, B, Cin, Cout, H, W, kernel, stride, dilation, padding, reflect, in_slope)
CONVS = [
    ('gen conv_pre k7', 16, 256, 512, 1, 40, (1, 7), (1, 1), (1, 1), (0, 3), False, 1.0),
    ('gen rb0 k11 d5 C256', 16, 256, 256, 1, 240, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1),
    ('gen rb1 k7 d3 C128', 16, 128, 128, 1, 1200, (1, 7), (1, 1), (1, 3), (0, 9), False, 0.1),
    ('gen rb2 k3 d1 C64', 16, 64, 64, 1, 6000, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
    ('gen rb3 k11 d1 C32', 8, 32, 32, 1, 12000, (1, 11), (1, 1), (1, 1), (0, 5), False, 0.1),
    ('gen conv_post k7 C32->1', 16, 32, 1, 1, 12000, (1, 7), (1, 1), (1, 1), (0, 3), False, 0.01),
    ('mpd p2 conv0 1->16', 16, 1, 16, 6000, 2, (5, 1), (3, 1), (1, 1), (2, 0), False, 1.0),
    ('mpd p3 conv1 16->64', 16, 16, 64, 1334, 3, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
    ('mpd p11 conv2 64->256', 16, 64, 256, 122, 11, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
    ('mpd p5 conv3 256->512', 16, 256, 512, 89, 5, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
    ('mpd p7 conv4 512->512 s1', 16, 512, 512, 22, 7, (5, 1), (1, 1), (1, 1), (2, 0), False, 0.2),
    ('mpd p2 post 512->1', 16, 512, 1, 75, 2, (3, 1), (1, 1), (1, 1), (1, 0), False, 0.2),
    ('mrd h15 conv0 2->4 s1', 16, 2, 4, 31, 801, (3, 3), (1, 1), (1, 1), (1, 1), True, 1.0),
    ('mrd h15 conv1 4->8 s2', 16, 4, 8, 31, 801, (3, 3), (2, 2), (1, 1), (1, 1), True, 1.0),
    ('mrd h240 conv3 64->128 s2', 16, 64, 128, 241, 26, (3, 3), (2, 2), (1, 1), (1, 1), True, 1.0),
    ('mrd h240 conv4 128->256 s1', 16, 128, 256, 121, 13, (3, 3), (1, 1), (1, 1), (1, 1), True, 1.0),
    ('mrd h240 conv6 512->1', 16, 512, 1, 61, 7, (3, 3), (1, 1), (1, 1), (1, 1), True, 1.0),
]


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-4), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize('case', CONVS, ids=[c[0] for c in CONVS])
def test_conv_forward_dgrad_wgrad(case, dtype, tol):
    from msmctts_amd.hip import conv
    name, B, Cin, Cout, H, W, k, s, dil, pad, reflect, slope = case
    g = torch.Generator(device='cpu').manual_seed(sum(ord(c) for c in name))
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(Cout, Cin, *k, generator=g) / (Cin * k[0] * k[1]) ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(Cout, generator=g).to(DEV).requires_grad_(True)
    xa = F.leaky_relu(x, slope) if slope != 1.0 else x
    if reflect:
        ref = F.conv2d(F.pad(xa, (pad[1], pad[1], pad[0], pad[0]), mode='reflect'), w, b, s, 0, dil)
    else:
        ref = F.conv2d(xa, w, b, s, pad, dil)
    go = torch.randn(ref.shape, generator=g).to(DEV)
    ref.backward(go)
    geom = conv.Geometry(H, W, k, s, dil, pad, reflect)
    T = k[0] * k[1]
    wf = w.detach().permute(2, 3, 0, 1).reshape(T, Cout, Cin).contiguous().to(dtype)
    wb = w.detach().permute(2, 3, 1, 0).reshape(T, Cin, Cout).contiguous().to(dtype)
    xc, gc = cl(x.detach()).to(dtype), cl(go).to(dtype)
    out = conv.conv_forward(xc, wf, geom, bias=b.detach(), in_slope=slope)
    assert rel(nchw(out), ref) < tol, 'forward'
    if reflect:
        gx = conv.reflect_fold(conv.conv_dgrad(gc, wb, geom), H, W, pad[0], mask_src=xc if slope != 1.0 else None,
                               slope=slope)
    else:
        gx = conv.conv_dgrad(gc, wb, geom, mask_src=xc if slope != 1.0 else None, mask_slope=slope)
    assert rel(nchw(gx), x.grad) < tol, 'dgrad'
    dw = conv.conv_wgrad(xc, gc, geom, T, in_slope=slope)
    want = w.grad.permute(2, 3, 0, 1).reshape(T, Cout, Cin)
    assert rel(dw, want) < max(tol, 1e-3 if dtype == torch.float32 else tol), 'wgrad'
    assert rel(conv.colsum(gc.reshape(-1, Cout)), b.grad) < max(tol, 1e-3), 'bias grad'


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-4), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize('Cin,Cout,k,u,L', [(512, 256, 12, 6, 40), (256, 128, 11, 5, 240), (128, 64, 11, 5, 1200),
                                            (64, 32, 4, 2, 6000)])
def test_conv_transpose(Cin, Cout, k, u, L, dtype, tol):
    from msmctts_amd.hip import conv
    B, p = 16, (k - u) // 2
    g = torch.Generator(device='cpu').manual_seed(Cin + k)
    x = torch.randn(B, Cin, L, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(Cin, Cout, k, generator=g) / (Cin * k / u) ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(Cout, generator=g).to(DEV)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, u, p)
    go = torch.randn(ref.shape, generator=g).to(DEV)
    ref.backward(go)
    xc = x.detach().permute(0, 2, 1).unsqueeze(1).contiguous().to(dtype)
    gc = go.permute(0, 2, 1).unsqueeze(1).contiguous().to(dtype)
    wf = w.detach().permute(2, 1, 0).contiguous().to(dtype)
    wb = w.detach().permute(2, 0, 1).contiguous().to(dtype)
    out = conv.conv_transpose1d_forward(xc, wf, k, u, p, bias=b, in_slope=0.1)
    assert rel(out.squeeze(1).permute(0, 2, 1), ref) < tol
    gx = conv.conv_transpose1d_dgrad(gc, wb, k, u, p, L, mask_src=xc, mask_slope=0.1)
    assert rel(gx.squeeze(1).permute(0, 2, 1), x.grad) < tol
    dw = conv.conv_transpose1d_wgrad(xc, gc, k, u, p, in_slope=0.1)
    assert rel(dw, w.grad.permute(2, 0, 1)) < max(tol, 1e-3)


def test_generator_and_discriminator_full_size_vs_oracle_ops():
    """CSMSC-size HifiGAN generator + discriminator through the HIP stacks (fp32) against the same
    network evaluated with PyTorch-ROCm convolutions (oracle.model functions on the GPU)."""
    from msmctts_amd.configs import csmsc_config
    from msmctts_amd.networks import find_modules
    from oracle.model import discriminator_forward, hifigan_generator
    cfg = csmsc_config()['task']
    torch.manual_seed(0)
    nets = dict(find_modules({k: v for k, v in cfg.items() if k[:1] != '_'}))
    gen, disc = nets['autoencoder'].decoder.to(DEV), nets['discriminator'].to(DEV)
    x = torch.randn(4, 256, 40, device=DEV)
    P = {'autoencoder.decoder.' + k: v.detach() for k, v in gen.state_dict().items()}
    P.update({'discriminator.' + k: v.detach() for k, v in disc.state_dict().items()})
    wav = gen(x)
    ref = hifigan_generator(P, 'autoencoder.decoder', x, cfg['autoencoder']['decoder_config'])
    assert (wav - ref).abs().max().item() < 1e-3
    scores, fmaps = disc(wav.detach())
    rs, rf = discriminator_forward(P, cfg['discriminator'], wav.detach())
    for a, b in zip(scores, rs):
        assert (a.float() - b).abs().max().item() < 1e-3
    for fa, fb in zip(fmaps, rf):
        for a, b in zip(fa, fb):
            assert a.shape == b.shape and (a.float() - b).abs().max().item() < 1e-3


@pytest.mark.parametrize('hop', [15, 30, 50, 120, 240])
def test_mrd_spectral_front_end_full_size(hop):
    """HIP framed-DFT front-end (B=16, L=12000) against the torch.stft-based oracle on the same device:
    image within 1e-4, gradient wrt the waveform within 1e-3 of its scale."""
    from msmctts_amd.utils.audio import TorchSTFT
    from oracle import audio
    g = torch.Generator().manual_seed(hop)
    x = (torch.rand(16, 12000, generator=g) * 2 - 1).to(DEV).requires_grad_(True)
    st = TorchSTFT(fft_size=4 * hop, hop_size=hop, win_size=4 * hop, normalized=True, domain='double', mel_scale=True)
    img = st.image_cl(x)
    xo = x.detach().clone().requires_grad_(True)
    ref = audio.mrd_spectrogram(xo, hop)                       # (B, 2, F, T)
    assert (img.permute(0, 3, 1, 2) - ref).abs().max().item() < 1e-4
    go = torch.randn(ref.shape, generator=g).to(DEV)
    ref.backward(go)
    img.backward(go.permute(0, 2, 3, 1).contiguous())
    assert (x.grad - xo.grad).abs().max().item() < 1e-3 * max(1.0, xo.grad.abs().max().item())


def test_mel_loss_full_size():
    from msmctts_amd.trainers.criterions.stft_loss import MelLoss
    from oracle import audio
    g = torch.Generator().manual_seed(1)
    a = (torch.rand(16, 12000, generator=g) * 2 - 1).to(DEV).requires_grad_(True)
    b = (torch.rand(16, 12000, generator=g) * 2 - 1).to(DEV)
    ml = MelLoss(fft_size=2048, hop_size=300, win_size=1200, sample_rate=24000, num_mels=128)
    loss = ml(a, b)
    ao = a.detach().clone().requires_grad_(True)
    ref = audio.mel_loss(ao, b)
    assert abs(loss.item() - ref.item()) < 1e-4
    loss.backward()
    ref.backward()
    assert (a.grad - ao.grad).abs().max().item() < 1e-3 * max(1e-6, ao.grad.abs().max().item()) + 1e-9
