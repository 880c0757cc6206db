"""Multi-head EMA vector quantiser ops over the gfx950 kernels (csrc/vq.hip).

Replaces the arithmetic of ``Quantize.forward`` / ``MultiHeadQuantize.forward``
(reference msmctts/networks/vqgantts/modules.py:24-67, :137-151).
"""
import os

import contextlib

import torch

from . import lib

# The shortlist search (csrc/vq_shortlist.inc: bf16 matrix-core shortlist + exact fp32 decision, bit-identical results)
# serves the shapes it takes once frames x codewords reaches SHORTLIST_MIN_WORK -- measured cross-over against the exact
# register-resident kernel on MI355X (profiles/r03_vq_shortlist.md: K = 64 from 131 072 frames, K = 256 from 32 768,
# K = 512 from 16 384; below it the eight-wave workgroups leave most of the chip idle and the exact kernel's smaller tiles
# win); below that, and for every other shape, the exact kernel runs (``SHORTLIST = False`` from a tool or test turns it off).
SHORTLIST = True
SHORTLIST_MIN_WORK = int(os.environ.get('MSMC_VQ_SHORTLIST_MIN_WORK', str(1 << 23)))
SLOW_COUNT = None           # tests / bench: an int64 [2] device tensor counting 16-frame tiles (per head) that took the
                            # two-candidate exact re-rank [0] / the full exact re-search [1]


def vq_prepare(embed, frames=None):
    """embed [H, d, K] -> (embed_t [H, K, d], enorm [H, K]); where the shortlist kernel takes the shape its codebook
    image rides along as ``embed_t.shortlist_image``.  ``frames``: the number of frames the caller is about to search
    (the modules pass it): below the cross-over the image would never be read and is not built (a launch per call)."""
    H, d, K = embed.shape
    embed_t = torch.empty((H, K, d), dtype=torch.float32, device=embed.device)
    enorm = torch.empty((H, K), dtype=torch.float32, device=embed.device)
    L = lib.get()
    lib.check(L.msmc_vq_prepare(lib.ptr(embed, torch.float32), lib.ptr(embed_t), lib.ptr(enorm), H, d, K,
                                lib.stream(embed)), 'msmc_vq_prepare')
    wanted = SHORTLIST and (frames is None or frames * K >= SHORTLIST_MIN_WORK)
    nbytes = int(L.msmc_vq_shortlist_bytes(H, d, K)) if wanted else 0
    if nbytes:
        image = torch.empty(nbytes, dtype=torch.uint8, device=embed.device)
        lib.check(L.msmc_vq_prepare_shortlist(lib.ptr(embed_t), lib.ptr(enorm), lib.ptr(image), H, d, K,
                                              lib.stream(embed)), 'msmc_vq_prepare_shortlist')
        embed_t.shortlist_image = image
    return embed_t, enorm


# the fp32 copy of the frames the last search ran on: in a bf16 step the search casts its input, and the EMA update that
# follows it (same frames, MultiHeadQuantize.forward) would cast them a second time
_F32_OF = {'last': None}


def _f32_frames(x):
    last = _F32_OF['last']
    if (last is not None and x.dtype != torch.float32 and last[0] == x.data_ptr() and last[1] == x.dtype
            and last[3].numel() == x.numel() and x.is_contiguous()):
        _F32_OF['last'] = None                # (one reader per search: nothing stale can be picked up by a later call)
        return last[3].view(x.shape)
    return x.detach().contiguous().float()


class _VQSearch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, embed_t, enorm, image=None, force_shortlist=False):
        H, K, d = embed_t.shape
        D = H * d
        assert x.shape[-1] == D, (x.shape, embed_t.shape)
        xc = x.contiguous().float()
        _F32_OF['last'] = (x.data_ptr(), x.dtype, tuple(x.shape), xc)      # (the EMA update of the same frames reads this copy)
        N = xc.numel() // D
        quant = torch.empty_like(xc)
        diff = torch.empty(xc.shape[:-1] + (d,), dtype=torch.float32, device=x.device)
        ind = torch.empty(xc.shape[:-1] + (H,), dtype=torch.int64, device=x.device)
        L = lib.get()
        if image is not None and (force_shortlist or N * K >= SHORTLIST_MIN_WORK):
            lib.check(L.msmc_vq_search_shortlist(lib.ptr(xc), lib.ptr(embed_t, torch.float32), lib.ptr(enorm, torch.float32),
                                                 lib.ptr(image, torch.uint8), lib.ptr(quant), lib.ptr(diff), lib.ptr(ind),
                                                 lib.ptr(SLOW_COUNT, torch.int64), N, D, H, K, lib.stream(xc)),
                      'msmc_vq_search_shortlist')
        else:
            lib.check(L.msmc_vq_search(lib.ptr(xc), lib.ptr(embed_t, torch.float32), lib.ptr(enorm, torch.float32),
                                       lib.ptr(quant), lib.ptr(diff), lib.ptr(ind), N, D, H, K, lib.stream(xc)),
                      'msmc_vq_search')
        ctx.save_for_backward(xc, quant)
        ctx.heads = H
        ctx.in_dtype = x.dtype
        ctx.mark_non_differentiable(ind)
        ctx.set_materialize_grads(False)      # (no zero-filled int64 "gradient" for the indices; quant / diff: handled below)
        return quant, diff, ind

    @staticmethod
    def backward(ctx, g_quant, g_diff, _g_ind):
        xc, quant = ctx.saved_tensors
        D = xc.shape[-1]
        N = xc.numel() // D
        if g_quant is None and g_diff is None:
            return None, None, None, None, None
        if g_quant is None:
            g_quant = torch.zeros_like(xc)
        g_quant = g_quant.contiguous().float()
        g_diff = None if g_diff is None else g_diff.contiguous().float()
        gx = torch.empty_like(xc)
        L = lib.get()
        lib.check(L.msmc_vq_backward(lib.ptr(g_quant), lib.ptr(g_diff), lib.ptr(xc), lib.ptr(quant), lib.ptr(gx),
                                     N, D, ctx.heads, lib.stream(xc)), 'msmc_vq_backward')
        return gx.to(ctx.in_dtype), None, None, None, None


def vq_search(x, embed_t, enorm, shortlist=None):
    """x [..., D] -> (quant [..., D] straight-through, diff [..., d], ind [..., H] int64).  ``shortlist``: None = the
    product's choice (the shortlist kernel where ``vq_prepare`` attached an image and the problem is large enough),
    True = the shortlist kernel whatever the size (it must have an image), False = the exact kernel."""
    image = getattr(embed_t, 'shortlist_image', None) if (SHORTLIST if shortlist is None else shortlist) else None
    if shortlist and image is None:
        raise RuntimeError('msmc_vq_search_shortlist does not take this shape (vq_prepare attached no shortlist image to this codebook)')
    return _VQSearch.apply(x, embed_t, enorm, image, bool(shortlist))


# The EMA update of a stage has no reader before the next step's search (the quantised values of this step came from the
# codebook as it was), yet its two launches (statistics 87 us + update 34 us per stage) sat in the middle of the autoencoder's
# forward chain.  Inside ``ema_side(device)`` -- MSMCVQGAN.forward around its quantiser -- they go to a side stream of the
# library's own (a parallel branch of the captured step, next to the frame decoder and the vocoder); ``join_ema`` at the end of
# that forward orders the caller behind them.  Stand-alone use of the quantiser modules keeps them on the calling stream.
EMA_SIDE = os.environ.get('MSMC_VQ_EMA_SIDE', '1') != '0'
_EMA = {'side': None, 'used': None}


@contextlib.contextmanager
def ema_side(device):
    from . import convnet
    keep = _EMA['side']
    if EMA_SIDE and torch.device(device).type == 'cuda' and convnet.STREAMS_ENABLED:
        _EMA['side'] = convnet.own_streams(device, 1, 'vq-ema')[0]
    try:
        yield
    finally:
        _EMA['side'] = keep


def join_ema(device):
    """the calling stream waits for the EMA updates issued inside ema_side since the last join"""
    st = _EMA['used']
    if st is not None:
        torch.cuda.current_stream(device).wait_stream(st)
        _EMA['used'] = None


def vq_ema_update(x, ind, length, embed, cluster_size, embed_avg, decay, eps, workspace=None):
    """In-place EMA update of the packed buffers embed [H,d,K], cluster_size [H,K], embed_avg [H,d,K]
    from the valid frames of x [B,T,D] / ind [B,T,H] (t < length[b])."""
    side = _EMA['side'] if x.is_cuda else None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream(x.device))
        _EMA['used'] = side
        with torch.cuda.stream(side):
            return _vq_ema_update(x, ind, length, embed, cluster_size, embed_avg, decay, eps, workspace)
    return _vq_ema_update(x, ind, length, embed, cluster_size, embed_avg, decay, eps, workspace)


def _vq_ema_update(x, ind, length, embed, cluster_size, embed_avg, decay, eps, workspace=None):
    B, T, D = x.shape
    H, d, K = embed.shape
    L = lib.get()
    need = int(L.msmc_vq_ema_workspace(B * T, D, H, K))
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
    xc = _f32_frames(x.detach())
    indc = ind.contiguous()                 # (named: a temporary would be released before the launch reads it)
    length = length.to(device=x.device, dtype=torch.int64).contiguous()
    lib.check(L.msmc_vq_ema_update(lib.ptr(xc), lib.ptr(indc, torch.int64), lib.ptr(length),
                                   lib.ptr(embed, torch.float32), lib.ptr(cluster_size, torch.float32),
                                   lib.ptr(embed_avg, torch.float32), lib.ptr(workspace),
                                   workspace.numel() * workspace.element_size(), B, T, D, H, K, float(decay),
                                   float(eps), lib.stream(xc)), 'msmc_vq_ema_update')
    return workspace


# -- data-parallel codebook synchronisation (opt-in; the reference's ranks EMA-update from their local batch) ----------
class CodebookSync(object):
    """Pending EMA statistics of one quantiser stage: this rank's [H][K][d] sums + [H][K] counts in a persistent buffer
    (static across hipGraph replays), applied to the packed buffers after the cross-rank sum."""

    def __init__(self, embed, cluster_size, embed_avg, decay, eps):
        H, d, K = embed.shape
        self.embed, self.cluster_size, self.embed_avg, self.decay, self.eps = embed, cluster_size, embed_avg, decay, eps
        self.stats = torch.zeros(H * K * (d + 2), dtype=torch.float32, device=embed.device)   # sums, counts, scratch
        self.workspace = None

    def matches(self, embed):
        return self.embed.data_ptr() == embed.data_ptr() and self.embed.shape == embed.shape

    def collect(self, x, ind, length):
        B, T, D = x.shape
        H, d, K = self.embed.shape
        L = lib.get()
        need = int(L.msmc_vq_ema_workspace(B * T, D, H, K))
        if self.workspace is None or self.workspace.numel() * 4 < need:
            self.workspace = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
        xc = x.detach().contiguous().float()
        indc = ind.contiguous()
        length = length.to(device=x.device, dtype=torch.int64).contiguous()
        lib.check(L.msmc_vq_ema_stats(lib.ptr(xc), lib.ptr(indc, torch.int64), lib.ptr(length), lib.ptr(self.stats),
                                      lib.ptr(self.workspace), self.workspace.numel() * 4, B, T, D, H, K, lib.stream(xc)),
                  'msmc_vq_ema_stats')

    def apply(self):
        H, d, K = self.embed.shape
        lib.check(lib.get().msmc_vq_ema_apply(lib.ptr(self.stats), lib.ptr(self.embed, torch.float32),
                                              lib.ptr(self.cluster_size, torch.float32), lib.ptr(self.embed_avg, torch.float32),
                                              H * d, H, K, float(self.decay), float(self.eps), lib.stream(self.stats)),
                  'msmc_vq_ema_apply')


_FLAT = {}            # (device, elements) -> persistent exchange buffer of flush_codebook_sync
PENDING = []          # CodebookSync objects whose statistics were collected by a forward and not yet applied


def flush_codebook_sync(pending=None, group=None, local=False):
    """ONE all-reduce (sum) over the statistics of every pending stage, then the per-stage buffer updates: afterwards
    every rank holds the codebooks a single process would have computed from the global batch.  ``local``: no exchange
    (the trainer's capture warm-up, which is rolled back)."""
    import torch.distributed as dist
    items = PENDING if pending is None else pending
    if not items:
        return
    if not local and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        # only the statistics travel: [H][K][d] sums + [H][K] counts of each stage (the trailing H*K words of a
        # stage's buffer are msmc_vq_ema_apply's scratch), through ONE persistent flat buffer
        views = [it.stats[:it.stats.numel() - it.embed.shape[0] * it.embed.shape[2]] for it in items]
        total = sum(v.numel() for v in views)
        key = (views[0].device, total)
        flat = _FLAT.get(key)
        if flat is None:
            # (buffers of other pending sets stay: a captured graph may still hold one -- round-3 advice; a handful of
            #  stage combinations exist per process)
            flat = _FLAT[key] = torch.empty(total, dtype=torch.float32, device=views[0].device)
        parts = list(flat.split([v.numel() for v in views]))
        torch._foreach_copy_(parts, views)
        dist.all_reduce(flat, group=group)
        torch._foreach_copy_(views, parts)
    for it in items:
        it.apply()
    if pending is None:
        del PENDING[:]
