"""Convolution layer cases shared by the GPU tests (full CSMSC sizes) and the kernel-interpreter tests
(same layer geometry, fewer pixels): every check compares the HIP kernels with PyTorch's own convolution."""
import torch
import torch.nn.functional as F


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

# the same geometries with few pixels / channels (the interpreter runs one work-item at a time), plus the
# thin and odd channel counts of the first discriminator layers
SMALL = [
    ('gen k11 d5 C64', 2, 64, 64, 1, 70, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1),
    ('gen k7 d3 C32', 2, 32, 32, 1, 150, (1, 7), (1, 1), (1, 3), (0, 9), False, 0.1),
    ('gen k3 C96->40', 1, 96, 40, 1, 40, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
    ('gen post C32->1', 2, 32, 1, 1, 140, (1, 7), (1, 1), (1, 1), (0, 3), False, 0.01),
    ('mpd 1->16', 2, 1, 16, 61, 2, (5, 1), (3, 1), (1, 1), (2, 0), False, 1.0),
    ('mpd 16->64 p3', 2, 16, 64, 40, 3, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
    ('mpd 64->1 p5', 1, 64, 1, 9, 5, (3, 1), (1, 1), (1, 1), (1, 0), False, 0.2),
    ('mrd 2->4 s1', 2, 2, 4, 9, 37, (3, 3), (1, 1), (1, 1), (1, 1), True, 1.0),
    ('mrd 4->8 s2', 2, 4, 8, 9, 37, (3, 3), (2, 2), (1, 1), (1, 1), True, 1.0),
    ('mrd 8->16 s1', 1, 8, 16, 11, 21, (3, 3), (1, 1), (1, 1), (1, 1), True, 1.0),
    ('mrd 64->72 s2', 1, 64, 72, 13, 18, (3, 3), (2, 2), (1, 1), (1, 1), True, 1.0),
    # inputs smaller than the receptive field / the reflection border
    ('tiny L3 k11 d5', 1, 32, 32, 1, 3, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1),
    ('tiny 2x2 reflect 4->8', 1, 4, 8, 2, 2, (3, 3), (1, 1), (1, 1), (1, 1), True, 1.0),
    ('tiny 3x2 reflect 64->64 s2', 2, 64, 64, 3, 2, (3, 3), (2, 2), (1, 1), (1, 1), True, 1.0),
]


def csmsc_layers(B=16):
    """Every convolution layer shape of the CSMSC model's HifiGAN generator, period and resolution discriminators
    (40-frame window = 12000 samples), as CONVS-style cases keyed by family."""
    out = {'gen': [], 'mpd': [], 'mrd': [], 'fft': []}
    # kernel-size-1 layers of the FFT blocks / quantiser glue (T = 400 and T / 4): Q|K|V and output projections, 1x1 stacks
    for T in (400, 100):
        for ci, co in ((256, 384), (128, 256), (256, 256), (512, 256), (80, 256), (256, 80)):
            out['fft'].append(('fft 1x1 T%d %d->%d' % (T, ci, co), B, ci, co, 1, T, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0))
    out['fft'].append(('fft ffn k3 256->1024', B, 256, 1024, 1, 400, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0))
    out['fft'].append(('fft ffn k3 1024->256 relu', B, 1024, 256, 1, 400, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.0))
    # the 600 / 1536-wide feed-forward layers of the predictor's FFT blocks (configs.am_config): channel counts that end inside a tile of 64
    out['fft'].append(('am ffn k3 600->1536', B, 600, 1536, 1, 100, (1, 3), (1, 1), (1, 1), (0, 1), False, 1.0))
    out['fft'].append(('am ffn k3 1536->600 relu', B, 1536, 600, 1, 100, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.0))
    out['gen'].append(('gen conv_pre k7 256->512', B, 256, 512, 1, 40, (1, 7), (1, 1), (1, 1), (0, 3), False, 1.0))
    for C, L in ((256, 240), (128, 1200), (64, 6000), (32, 12000)):
        for k in (3, 7, 11):
            for d in (1, 3, 5):
                out['gen'].append(('gen rb C%d L%d k%d d%d' % (C, L, k, d), B, C, C, 1, L, (1, k), (1, 1), (1, d),
                                   (0, d * (k - 1) // 2), False, 0.1))
    out['gen'].append(('gen conv_post k7 32->1', B, 32, 1, 1, 12000, (1, 7), (1, 1), (1, 1), (0, 3), False, 0.01))
    for p in (2, 3, 5, 7, 11):
        H = -(-12000 // p)
        chans = (1, 16, 64, 256, 512)
        for i in range(4):
            out['mpd'].append(('mpd p%d conv%d %d->%d s3' % (p, i, chans[i], chans[i + 1]), B, chans[i], chans[i + 1], H, p,
                               (5, 1), (3, 1), (1, 1), (2, 0), False, 1.0 if i == 0 else 0.2))
            H = (H + 4 - 5) // 3 + 1
        out['mpd'].append(('mpd p%d conv4 512->512 s1' % p, B, 512, 512, H, p, (5, 1), (1, 1), (1, 1), (2, 0), False, 0.2))
        out['mpd'].append(('mpd p%d post 512->1' % p, B, 512, 1, H, p, (3, 1), (1, 1), (1, 1), (1, 0), False, 0.2))
    for hop, hidden in ((15, 128), (30, 128), (50, 256), (120, 256), (240, 512)):
        F_, T_ = 2 * hop + 1, 12000 // hop + 1
        chans = (2, hidden // 32, hidden // 16, hidden // 8, hidden // 4, hidden // 2, hidden, 1)
        for i, st in enumerate((1, 2, 1, 2, 1, 2, 1)):
            out['mrd'].append(('mrd h%d conv%d %d->%d s%d' % (hop, i, chans[i], chans[i + 1], st), B, chans[i], chans[i + 1],
                               F_, T_, (3, 3), (st, st), (1, 1), (1, 1), True, 1.0 if i == 0 else 0.2))
            F_, T_ = (F_ + 2 - 3) // st + 1, (T_ + 2 - 3) // st + 1
    return out


def conv_case_data(case, dtype, dev):
    """operands in the kernels' layouts and PyTorch's results (forward, data gradient, weight / bias gradient) of one
    layer: computed once per (case, dtype), checked against as many kernel variants as the caller forces"""
    name, B, Cin, Cout, H, W, k, s, dil, pad, reflect, slope = case
    g = torch.Generator(device='cpu').manual_seed(sum(ord(c) for c in name))
    x = torch.randn(B, Cin, H, W, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(Cout, Cin, *k, generator=g) / (Cin * k[0] * k[1]) ** 0.5).to(dev).requires_grad_(True)
    b = torch.randn(Cout, generator=g).to(dev).requires_grad_(True)
    xa = F.leaky_relu(x, slope) if slope != 1.0 else x
    if reflect:
        ref = F.conv2d(F.pad(xa, (pad[1], pad[1], pad[0], pad[0]), mode='reflect'), w, b, s, 0, dil)
    else:
        ref = F.conv2d(xa, w, b, s, pad, dil)
    go = torch.randn(ref.shape, generator=g).to(dev)
    ref.backward(go)
    T = k[0] * k[1]
    d = dict(case=case, dtype=dtype, T=T, ref=ref.detach(), gx_ref=x.grad, b=b.detach(),
             dw_ref=w.grad.permute(2, 3, 0, 1).reshape(T, Cout, Cin),
             wf=w.detach().permute(2, 3, 0, 1).reshape(T, Cout, Cin).contiguous().to(dtype),
             wb=w.detach().permute(2, 3, 1, 0).reshape(T, Cin, Cout).contiguous().to(dtype),
             xc=cl(x.detach()).to(dtype), gc=cl(go).to(dtype))
    d['db_ref'] = d['gc'].float().reshape(-1, Cout).sum(0)
    d['db_scale'] = d['gc'].float().reshape(-1, Cout).abs().sum(0).max().item()
    return d


def conv_part_errors(d, part):
    """relative error(s) of one part ('fwd' / 'dgrad' / 'wgrad') of a prepared case under the CURRENT kernel choices
    (a fresh Geometry per call: descriptors, and with them forced variants, are cached per geometry object)"""
    from msmctts_amd.hip import conv
    name, B, Cin, Cout, H, W, k, s, dil, pad, reflect, slope = d['case']
    geom = conv.Geometry(H, W, k, s, dil, pad, reflect)
    if part == 'fwd':
        out = conv.conv_forward(d['xc'], d['wf'], geom, bias=d['b'], in_slope=slope)
        return {'fwd': rel(nchw(out), d['ref'])}
    if part == 'dgrad':
        if reflect:
            gx = conv.reflect_fold(conv.conv_dgrad(d['gc'], d['wb'], geom), H, W, pad[0],
                                   mask_src=d['xc'] if slope != 1.0 else None, slope=slope)
        else:
            gx = conv.conv_dgrad(d['gc'], d['wb'], geom, mask_src=d['xc'] if slope != 1.0 else None, mask_slope=slope)
        return {'dgrad': rel(nchw(gx), d['gx_ref'])}
    db = torch.zeros(Cout, device=d['xc'].device)
    dw = conv.conv_wgrad(d['xc'], d['gc'], geom, d['T'], in_slope=slope, db=db)
    return {'wgrad': rel(dw, d['dw_ref']), 'bias grad': (db - d['db_ref']).abs().max().item() / (d['db_scale'] + 1e-6) * 1e3}


def check_conv_case(case, dtype, tol, dev, parts=('fwd', 'dgrad', 'wgrad'), batch_offset=0):
    """forward / data gradient / weight + bias gradient of one layer against PyTorch on ``dev``."""
    from msmctts_amd.hip import conv
    name, B, Cin, Cout, H, W, k, s, dil, pad, reflect, slope = case
    g = torch.Generator(device='cpu').manual_seed(sum(ord(c) for c in name))
    x = torch.randn(B, Cin, H, W, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(Cout, Cin, *k, generator=g) / (Cin * k[0] * k[1]) ** 0.5).to(dev).requires_grad_(True)
    b = torch.randn(Cout, generator=g).to(dev).requires_grad_(True)
    xa = F.leaky_relu(x, slope) if slope != 1.0 else x
    if reflect:
        ref = F.conv2d(F.pad(xa, (pad[1], pad[1], pad[0], pad[0]), mode='reflect'), w, b, s, 0, dil)
    else:
        ref = F.conv2d(xa, w, b, s, pad, dil)
    go = torch.randn(ref.shape, generator=g).to(dev)
    ref.backward(go)
    geom = conv.Geometry(H, W, k, s, dil, pad, reflect)
    T = k[0] * k[1]
    wf = w.detach().permute(2, 3, 0, 1).reshape(T, Cout, Cin).contiguous().to(dtype)
    wb = w.detach().permute(2, 3, 1, 0).reshape(T, Cin, Cout).contiguous().to(dtype)
    xc, gc = cl(x.detach()).to(dtype), cl(go).to(dtype)
    if batch_offset:
        # operands that start in the middle of a larger allocation at an element-aligned address are refused
        # loudly (the kernels use 16-byte vector accesses), not mis-read
        import pytest
        buf = torch.empty(xc.numel() + batch_offset, dtype=xc.dtype, device=xc.device)
        xo = buf[batch_offset:].view(xc.shape)
        xo.copy_(xc)
        with pytest.raises(ValueError, match='16-byte aligned'):
            conv.conv_forward(xo, wf, geom, bias=b.detach(), in_slope=slope)
        with pytest.raises(ValueError, match='16-byte aligned'):
            conv.conv_wgrad(xo, gc, geom, T, in_slope=slope)
    if 'fwd' in parts:
        out = conv.conv_forward(xc, wf, geom, bias=b.detach(), in_slope=slope)
        assert rel(nchw(out), ref) < tol, 'forward'
    if 'dgrad' in parts:
        if reflect:
            gx = conv.reflect_fold(conv.conv_dgrad(gc, wb, geom), H, W, pad[0],
                                   mask_src=xc if slope != 1.0 else None, slope=slope)
        else:
            gx = conv.conv_dgrad(gc, wb, geom, mask_src=xc if slope != 1.0 else None, mask_slope=slope)
        assert rel(nchw(gx), x.grad) < tol, 'dgrad'
    if 'wgrad' in parts:
        db = torch.zeros(Cout, device=dev)
        dw = conv.conv_wgrad(xc, gc, geom, T, in_slope=slope, db=db)
        want = w.grad.permute(2, 3, 0, 1).reshape(T, Cout, Cin)
        wtol = max(tol, 1e-3 if dtype == torch.float32 else tol)
        assert rel(dw, want) < wtol, ('wgrad', rel(dw, want))
        # the fused bias gradient sums exactly the (rounded) operand the kernel was given
        bref = gc.float().reshape(-1, Cout).sum(0)
        bscale = gc.float().reshape(-1, Cout).abs().sum(0).max().item()
        assert (db - bref).abs().max().item() < 1e-5 * bscale + 1e-6, ('fused bias grad', db, bref)
        # privatised accumulators: R copies back to back, workgroup i adds into copy i % R; their sum is dW
        R = 4
        dwR = torch.zeros(R, T, Cout, Cin, device=dev)
        dbR = torch.zeros(R, Cout, device=dev)
        conv.conv_wgrad(xc, gc, geom, T, in_slope=slope, dw=dwR.view(-1), db=dbR.view(-1), copies=R)
        assert rel(dwR.sum(0), want) < wtol, ('wgrad copies', rel(dwR.sum(0), want))
        assert (dbR.sum(0) - bref).abs().max().item() < 1e-5 * bscale + 1e-6, 'bias grad copies'
        cs = conv.colsum(gc.reshape(-1, Cout))
        assert (cs - bref).abs().max().item() < 1e-5 * bscale + 1e-6, ('bias grad', cs, bref)
