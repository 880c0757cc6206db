"""Weight-normalised convolution stacks on the gfx950 implicit-GEMM kernels.

``ConvBank`` owns, for all weight-normalised convolutions of one top-level network (HifiGAN generator,
UnivNet discriminator):
  * the kernel-layout weights (forward slices ``[T, Cout, Cin]`` and data-gradient slices
    ``[T, Cin, Cout]`` in the compute dtype), refreshed from ``weight_v / weight_g`` by ONE
    ``msmc_wn_prepare_multi_tiles`` call (two launches) per forward pass of the network;
  * fp32 weight-gradient accumulators in kernel layout, filled by ``msmc_conv_wgrad`` from each
    convolution's backward, and turned into ``weight_v.grad / weight_g.grad / bias.grad`` by ONE
    ``msmc_wn_backward_multi`` launch at the end of the backward pass (autograd engine callback).
``hip_conv`` is the autograd function of one fused convolution
    out = lrelu_out( (res2 + ((conv(lrelu_in(x)) + bias) + res)) / div ).
Activations are channels-last ``[B, H, W, C]`` (1-D signals: H == 1), float32 or bfloat16.
"""
import contextlib
import ctypes

import os
import weakref

import torch
from torch.autograd import Variable

from . import conv as K
from . import lib


# Set by msmctts_amd.distributed.distributed.apply_gradient_allreduce: callable(param) telling the
# data-parallel reducer that a parameter gradient produced outside autograd's accumulation is ready.
GRAD_READY_HOOK = None

DW_COPIES = 8
DW_COPIES_MAX_ELEMS = 256 * 1024

# False: fork_join runs its branches back to back on the calling stream (bench.py's per-kernel timing pass)
STREAMS_ENABLED = True

# Weight gradients of convolutions that are NOT part of a grouped call (the FFT blocks' projections and feed-forward
# layers, the pre- / post-nets): each alone is a grid of 24-100 workgroups on 256 CUs, and none of them has a consumer
# before the end of the backward pass.  They wait in the bank until WGRAD_BATCH of them are pending (or the pass ends)
# and go out as ONE grouped call (K.conv_wgrad_group; the tuner keeps separate launches where grouping does not pay).
WGRAD_BATCH = int(os.environ.get('MSMC_WGRAD_BATCH', '8'))
WGRAD_BATCH_GROUPS = os.environ.get('MSMC_WGRAD_BATCH_GROUPS', '1') == '1'    # grouped calls' members join the waiting list too
# 1: a bank delivers its parameter gradients as soon as its last backward node has run (see ConvBank._open_nodes)
EARLY_FINISH = os.environ.get('MSMC_EARLY_FINISH', '1') != '0'
FINISH_SIDE = os.environ.get('MSMC_FINISH_SIDE', '1') != '0'     # ... and does so on a side stream (see ConvBank.node_closed)


def fork_join(streams, thunks, inputs=(), main_thunk=None):
    """Run independent launch sequences on side HIP streams and join them back (hipGraph-capturable).
    ``main_thunk`` (optional) runs on the calling stream between the fork and the join; its result comes last.

    The sub-discriminators / parallel ResBlocks are chains of small kernels (tens of workgroups): run back to
    back they leave most of the 256 CUs idle, concurrently they fill the chip.  ``inputs`` are tensors produced
    on the calling stream and read by the side streams; every tensor returned by a thunk is produced on a side
    stream and consumed by the caller -- both directions are registered with the caching allocator.
    Autograd replays each backward node on the stream of its forward, so the backward pass forks the same way.
    """
    if not streams or not STREAMS_ENABLED:
        return [t() for t in thunks] + ([main_thunk()] if main_thunk is not None else [])
    main = torch.cuda.current_stream()
    outs = []
    for i, thunk in enumerate(thunks):
        st = streams[i % len(streams)]
        st.wait_stream(main)
        for x in inputs:
            x.record_stream(st)
        with torch.cuda.stream(st):
            outs.append(thunk())
    tail = [main_thunk()] if main_thunk is not None else []
    for st in streams[:len(thunks)]:
        main.wait_stream(st)

    def mark(o):
        if torch.is_tensor(o):
            o.record_stream(main)
        elif isinstance(o, (list, tuple)):
            for v in o:
                mark(v)
    mark(outs)
    return outs + tail


# Weight gradients are off the critical path of a backward pass (only the optimizer reads them): MSMC_WGRAD_STREAMS=n > 0
# issues them round-robin on n side streams -- parallel branches of the captured hipGraph -- while the data-gradient chain
# continues on the calling stream.  Branches of a hipGraph DO run concurrently on this runtime (tools/graph_fork_probe.py: two
# chains of GEMM launches replay 1.5-1.9x faster forked than serial; chains of streaming elementwise launches gain nothing).
# History: at 29-31 ms/step (round 2) the gain was within noise -- every launch then occupied all 256 CUs for long enough
# that a second branch only got the tails.  At 17.6 ms/step (round 4; kernels 2x shorter, so the tails and the launch
# boundaries weigh more) it is measurable: n = 0 / 1 / 2 / 4 / 8 -> 17.62 / 17.60 / 17.12 / 17.02 / 17.05 ms.
# n = 1 buys nothing: a weight gradient next to the data-gradient chain fills the same LDS-bound slots; the gain is the
# weight gradients of DIFFERENT layers (partial grids, split-K tails) overlapping each other.
# The discriminator's resolution and period families are two more branches (networks/hifigan/discriminator.py D_FORK): another
# 0.4 ms.  All of these are streams of the library's own (own_streams below), not torch pool streams.
WGRAD_STREAMS = int(os.environ.get('MSMC_WGRAD_STREAMS', '12'))
_SIDE = {}
_OWN = {}


def own_streams(device, n, role):
    """n HIP streams of the library's own (msmc_stream_create) wrapped for torch: created once per process, device and
    ``role`` (every user of a role shares them -- one step runs at a time) and never returned.  torch.cuda.Stream() draws from
    a pool of 32 per device and starts over after the 32nd: in a process that builds several trainers, a 'side' stream can then
    BE the stream a later capture runs on, or another side stream, and a fork onto it is no branch at all (a capture whose
    branches were pool streams crashed the runtime at capture_end after a few trainers in one process)."""
    device = torch.device(device)
    have = _OWN.setdefault((device, role), [])
    with torch.cuda.device(device):
        while len(have) < n:
            h = ctypes.c_void_p()
            lib.check(lib.get().msmc_stream_create(ctypes.byref(h)), 'msmc_stream_create')
            have.append(torch.cuda.ExternalStream(h.value, device=device))
    return have[:n]


def _side_streams(device):
    st = _SIDE.get(device)
    if st is None:
        st = _SIDE[device] = own_streams(device, max(1, WGRAD_STREAMS), 'wgrad')
    return st


def make_streams(device, n):
    """n side streams on a CUDA/HIP device; [] on the CPU (kernel-interpreter tests run sequentially)."""
    if device.type != 'cuda' or n <= 1 or os.environ.get('MSMC_STREAMS', '1') == '0' or GROUPED:
        return []
    return [torch.cuda.Stream(device=device) for _ in range(n)]


class ConvLayer(object):
    """Static description of one convolution inside a bank: weight-normalised (``weight_g`` / ``weight_v``) or, with
    ``plain=True``, an ordinary ``nn.Conv1d`` whose ``weight`` is used as it is."""

    def __init__(self, module, kind, kernel, stride=(1, 1), dilation=(1, 1), padding=(0, 0), reflect=False,
                 plain=False):
        self.module = module            # owns bias / weight_g / weight_v (or bias / weight) parameters
        self.plain = plain
        self.kind = kind                # 'conv' (Conv1d as (1,k) / Conv2d) or 'convT' (ConvTranspose1d)
        self.kernel, self.stride, self.dilation, self.padding, self.reflect = kernel, stride, dilation, padding, reflect
        self.taps = kernel[0] * kernel[1]
        v = self.weight
        if kind == 'conv':
            self.cout, self.cin = v.shape[0], v.shape[1]
        else:
            self.cin, self.cout = v.shape[0], v.shape[1]
        # filled by the bank
        self.wf = self.wb = self.dw = self.db = None
        self.wg_variants = set()        # msmc_conv_wgrad variants this layer's weight gradients have run on (see ConvBank._drop_idle_copies)
        self.index = -1
        self._geoms = {}

    @property
    def weight(self):
        """the parameter the kernel-layout weights are derived from (weight_v under weight norm)"""
        return self.module.weight if self.plain else self.module.weight_v

    def geom(self, H, W):
        key = (H, W)
        g = self._geoms.get(key)
        if g is None:
            K._bounded(self._geoms, 1024)         # one entry per input extent: bounded under variable-length batches
            g = self._geoms[key] = K.Geometry(H, W, self.kernel, self.stride, self.dilation, self.padding, self.reflect)
        return g


# The kernel-layout weight images of a bank only change when its parameters do.  A step runs the discriminator three times
# (D step, then twice against the updated D) and the next step's first pass still sees the weights of the previous step's
# last two: ``prepare`` launches nothing while the bank is CLEAN -- no optimizer step of its parameters since the last
# refresh (``params_updated``, called by HipAdamW, whose raw-pointer update torch's version counters cannot see) and no
# in-place torch operation on them either (their ``_version`` counters: load_state_dict, init, the capture roll-back).
# MSMC_SKIP_CLEAN_PREPARE=0 refreshes on every forward pass (A/B).
SKIP_CLEAN_PREPARE = os.environ.get('MSMC_SKIP_CLEAN_PREPARE', '1') != '0'
_BANKS = weakref.WeakSet()


def params_updated(owner, params):
    """an optimizer wrote ``params`` through raw pointers: the banks holding any of them must refresh their weight images"""
    ptrs = getattr(owner, '_msmc_ptrs', None)
    if ptrs is None or ptrs[0] != len(params):
        ptrs = owner._msmc_ptrs = (len(params), frozenset(p.data_ptr() for p in params))
    for bank in list(_BANKS):
        hit = bank._owner_hits.get(id(owner))
        if hit is None or hit[0] is not ptrs:
            hit = bank._owner_hits[id(owner)] = (ptrs, bool(bank.param_ptrs() & ptrs[1]))
        if hit[1]:
            bank.dirty = True


def graphs_replayed():
    """a captured optimizer segment was replayed: the parameters moved, and so did the weight images of the banks the graphs
    refresh -- but neither the ``dirty`` flags nor torch's version counters saw it (the update and the refresh are graph
    nodes).  The graphs keep themselves consistent; an EAGER forward in the same process (validation, synthesis, the bench's
    instrumented steps) must refresh before it computes: every built bank is marked for that."""
    for bank in list(_BANKS):
        if bank._sig is not None:
            bank.eager_stale = True


def refresh_stale_banks():
    """eager refresh of every bank whose parameters changed behind a captured step's back (checkpoint load between replays,
    the roll-back of the capture warm-up): the replayed graphs only refresh a bank where the capture saw it dirty"""
    for bank in list(_BANKS):
        if bank._sig is not None and (bank.dirty or bank._clean_versions != bank._versions()):
            bank.prepare(bank.dtype)


# A model is several banks (the autoencoder: in / out projections, two encoder stacks, quantiser, frame decoder, vocoder) whose
# parameters ONE optimizer step moves together, so at the head of the next forward pass all of them are stale -- and each
# refreshed its images when its module was reached: six calls of two launches, 12 nodes and 0.26 ms on the critical chain of
# the step (profiles/r06_step_timeline_start_of_round.txt).  ``prepare_together`` refreshes every stale bank of a list in
# ONE msmc_wn_prepare_multi_tiles call over the concatenation of their item tables (block offsets re-based); the banks' own
# ``prepare`` calls further down then find themselves clean.  MSMC_PREPARE_TOGETHER=0: each bank on its own (A/B).
PREPARE_TOGETHER = os.environ.get('MSMC_PREPARE_TOGETHER', '1') != '0'
_TOGETHER = {}


def prepare_together(pairs):
    """``pairs``: [(bank, dtype), ...] -- banks about to be used by one forward pass.  No effect on numerics or on what a
    bank considers clean; fewer than two stale banks are left to their own ``prepare``."""
    if not PREPARE_TOGETHER:
        return
    stale = []
    for bank, dtype in pairs:
        need, versions, capturing = bank._stale(dtype)
        if need:
            stale.append((bank, versions, capturing))
    if len(stale) < 2:
        return
    banks = [b for b, _, _ in stale]
    # (keyed by the banks' builds: the fields the two prepare kernels read -- pointers, extents, strides, block offsets -- change
    #  with a rebuild only; ``copies``, which _drop_idle_copies patches later, is the backward pass's)
    key = tuple((id(b), b._build_version) for b in banks)
    hit = _TOGETHER.get(key)
    if hit is None:
        K._bounded(_TOGETHER, 64)
        n = sum(len(b.layers) for b in banks)
        items = (lib.WnItem * n)()
        k = blk = tblk = 0
        for b in banks:
            for it in b._items_host:
                ctypes.memmove(ctypes.byref(items[k]), ctypes.byref(it), ctypes.sizeof(lib.WnItem))
                items[k].block0 = it.block0 + blk
                items[k].tblock0 = it.tblock0 + tblk
                k += 1
            blk += b.total_blocks
            tblk += b.total_tile_blocks
        dev = banks[0].w1.device
        if dev.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            return                      # (a table built during capture would be a host-to-device copy node: banks one by one)
        table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(dev)
        hit = _TOGETHER[key] = (table, n, blk, tblk, max(b.max_taps for b in banks), _wn_maps(items, dev),
                                sum(b.w1.numel() for b in banks), banks[0].w1.element_size())
    table, n, blk, tblk, max_taps, maps = hit[:6]
    _wn_prepare(table, maps, n, blk, tblk, max_taps, lib.stream(banks[0].w1), 'msmc_wn_prepare_multi_tiles(together)')
    for bank, versions, capturing in stale:
        bank._refreshed(versions, capturing)


def _wn_maps(items, device):
    """device index maps of msmc_wn_prepare_multi_tiles over a host item table: (int32 tensor, offset of norm_rows, number of
    them, offset of tile_item) -- row_item first.  Block k of the norms pass / the layout pass reads its item from these
    instead of searching the item table (eight dependent loads per workgroup)."""
    L = lib.get()
    row_item, norm_rows, tile_item = [], [], []
    for i, it in enumerate(items):
        row_item += [i] * it.A
        if it.g:
            norm_rows += range(it.block0, it.block0 + it.A)
        tile_item += [i] * int(L.msmc_wn_tile_blocks(it.A, it.Bc, it.T))
    flat = torch.tensor(row_item + norm_rows + tile_item + [0], dtype=torch.int32).to(device)
    return flat, len(row_item), len(norm_rows), len(row_item) + len(norm_rows)


def _wn_prepare(table, maps, n, blk, tblk, max_taps, stream, what):
    flat, o_norm, n_norm, o_tile = maps
    base = flat.data_ptr()
    lib.check(lib.get().msmc_wn_prepare_multi_tiles(lib.ptr(table), n, blk, tblk, max_taps, ctypes.c_void_p(base),
                                                    ctypes.c_void_p(base + 4 * o_norm), n_norm,
                                                    ctypes.c_void_p(base + 4 * o_tile), stream), what)


class ConvBank(object):
    def __init__(self, layers):
        self.layers = list(layers)
        # autograd nodes of this bank whose backward (with a weight gradient) is still to come in the running pass: when the
        # count returns to zero the bank's gradients are complete and ``_finish_backward`` runs AT ONCE -- not at the end of
        # the whole backward pass -- so that the data-parallel reducer can send the generator's gradients while the frame
        # decoder / quantiser / encoders are still back-propagating (EARLY_FINISH; the end-of-pass callback stays as the
        # net for passes the count cannot see through: outputs nobody used, exceptions)
        self._open_nodes = 0
        self._close_task, self._close_mixed = None, False      # graph task of the nodes closed so far / more than one seen
        self.dirty = True               # kernel-layout weights older than the parameters (see SKIP_CLEAN_PREPARE)
        self.eager_stale = False        # ... as far as an EAGER pass can tell: graph replays moved the parameters (graphs_replayed)
        self._clean_versions = None
        self._owner_hits = {}
        self._ptrs = None
        _BANKS.add(self)
        for i, l in enumerate(self.layers):
            l.index = i
        self._sig = None
        self._queued = False
        self._touched = set()           # layers whose weight gradient was accumulated in the running backward pass
        self._hold = []                 # gradients shared by several consumers, pinned until the backward ends
        self.streams = []               # side streams whose backward launches write this bank's accumulators
        self._side_used = []            # weight-gradient side streams to join at the end of the backward pass
        self._side_rr = 0
        self.deferred = K.DeferredReduce()     # partial-result arena + pending second stages of the running backward pass
        self._pending_w = {}            # stream -> (Stream, weight gradients waiting for company), see WGRAD_BATCH
        self._dw_stream = {}            # accumulator -> side stream of its last launch in the running pass (wgrad_side)
        self._finish_stream = None      # side stream an early delivery of the running pass was issued on (node_closed)

    def queue_wgrad(self, item):
        """Backward nodes replay on the stream of their forward (fork_join branches): a waiting list per stream, flushed
        on that stream, so that a grouped launch only reads what its own stream produced."""
        st = torch.cuda.current_stream(item['x'].device) if item['x'].is_cuda else None
        key = st.cuda_stream if st is not None else 0
        items = self._pending_w.setdefault(key, (st, []))[1]
        if any(it['dw'].data_ptr() == item['dw'].data_ptr() for it in items):
            self._flush_stream(key)     # a layer applied twice: its two accumulations must not share a launch
            items = self._pending_w.setdefault(key, (st, []))[1]
        items.append(item)
        if len(items) >= WGRAD_BATCH:
            self._flush_stream(key)

    def _flush_stream(self, key):
        st, items = self._pending_w.pop(key, (None, []))
        if not items:
            return
        with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
            with self.wgrad_side(*([it['x'] for it in items] + [it['g'] for it in items]), dws=[it['dw'] for it in items]):
                K.conv_wgrad_group(items)

    def flush_wgrad(self):
        for key in list(self._pending_w):
            self._flush_stream(key)

    def _order_accumulators(self, st, dws):
        for dw in dws:
            prev = self._dw_stream.get(dw.data_ptr())
            if prev is not None and prev.cuda_stream != st.cuda_stream:
                st.wait_stream(prev)
            self._dw_stream[dw.data_ptr()] = st

    @contextlib.contextmanager
    def wgrad_side(self, *tensors, dws=()):
        """Context for the weight-gradient launches of one backward node: a side stream ordered after everything the
        calling stream has issued so far.  ``tensors`` (the node's inputs to those launches) stay referenced until the
        join in ``_finish_backward``, so the caching allocator -- eagerly and inside a capture -- cannot hand their
        memory to a later allocation of the calling stream while the side stream still reads it.  ``dws``: the accumulators
        the launches add to -- a layer applied twice in one pass (the two batch halves of a forked block stack) must not have
        its two accumulations in flight at once (a single-split launch adds to dW without atomics), so the stream chosen
        here first waits for the stream that ran the previous launch on any of them.
        (Tried on top of this: a block stack running the two halves of its batch as two branches -- FFT stacks, 50-400
        workgroups per launch.  16.27 -> 18.44 ms/step: twice the launches at nearly the same duration each, and the
        branches did not overlap enough to pay for them.  Not kept.)"""
        dev = self.w1.device
        if dev.type != 'cuda' or WGRAD_STREAMS <= 0 or not STREAMS_ENABLED:
            if dev.type == 'cuda':              # (launches on the calling stream -- which may be one of two branches, see dws)
                self._order_accumulators(torch.cuda.current_stream(dev), dws)
            K.DEFER_TO = self.deferred          # (second stages of these launches: once, in _finish_backward)
            try:
                yield
            finally:
                K.DEFER_TO = None
            return
        sts = _side_streams(dev)
        st = sts[self._side_rr % len(sts)]
        self._side_rr += 1
        st.wait_stream(torch.cuda.current_stream(dev))
        self._order_accumulators(st, dws)
        self._hold.extend(t for t in tensors if t is not None)
        if not any(st is u for u in self._side_used):
            self._side_used.append(st)
        with torch.cuda.stream(st):
            K.DEFER_TO = self.deferred
            try:
                yield
            finally:
                K.DEFER_TO = None

    # -- (re)build device buffers whenever parameters moved / changed dtype ------------------------
    def _signature(self, dtype):
        return (dtype,) + tuple(l.weight.data_ptr() for l in self.layers)

    def _build(self, dtype):
        dev = self.layers[0].weight.device
        pad8 = lambda n: (n + 7) // 8 * 8                 # every layer's slices start 16-byte aligned (vector loads)
        tot_w = sum(pad8(l.taps * l.cout * l.cin) for l in self.layers)
        tot_a = sum(l.weight.shape[0] for l in self.layers)
        tot_b = sum(l.cout for l in self.layers)
        # privatised dW / db accumulators: atomics on one address retire serially (~0.1 us each), so layers with
        # small weights (= many pixel tiles per weight) spread their workgroups over DW_COPIES copies
        copies = lambda l: DW_COPIES if l.taps * l.cout * l.cin <= DW_COPIES_MAX_ELEMS else 1
        tot_dw = sum(pad8(l.taps * l.cout * l.cin) * copies(l) for l in self.layers)
        tot_db = sum(pad8(l.cout) * copies(l) for l in self.layers)
        self.w1 = torch.empty(tot_w, dtype=dtype, device=dev)            # layout 1 (the layout dW is produced in)
        self.w2 = torch.empty(tot_w, dtype=dtype, device=dev)            # layout 2
        self.dw = torch.zeros(tot_dw, dtype=torch.float32, device=dev)
        self.gv = torch.empty(tot_w, dtype=torch.float32, device=dev)
        self.inv_norm = torch.empty(tot_a, dtype=torch.float32, device=dev)
        self.gg = torch.empty(tot_a, dtype=torch.float32, device=dev)
        self.db = torch.zeros(tot_db, dtype=torch.float32, device=dev)
        self.gb = torch.empty(tot_b, dtype=torch.float32, device=dev)
        items = (lib.WnItem * len(self.layers))()
        ow = oa = ob = blk = odw = odb = tblk = 0
        esz = self.w1.element_size()
        for l, it in zip(self.layers, items):
            v, g = l.weight, (None if l.plain else l.module.weight_g)
            n = l.taps * l.cout * l.cin
            A = v.shape[0]
            it.v, it.g = v.data_ptr(), (g.data_ptr() if g is not None else None)
            it.dst1, it.dst2 = self.w1.data_ptr() + ow * esz, self.w2.data_ptr() + ow * esz
            it.inv_norm = self.inv_norm.data_ptr() + oa * 4
            R = copies(l)
            it.dw, it.gv, it.gg = self.dw.data_ptr() + odw * 4, self.gv.data_ptr() + ow * 4, self.gg.data_ptr() + oa * 4
            it.copies, it.dw_copy_stride, it.db_copy_stride = R, n, l.cout
            l.dw_copies = R
            it.A, it.Bc, it.T = A, v.shape[1], l.taps
            it.dtype = 0 if dtype == torch.float32 else 1
            it.block0 = blk
            it.tblock0 = tblk                      # tiles (rows x columns x all taps) of the layout pass, csrc/conv.hip wn_layout_kernel
            tblk += int(lib.get().msmc_wn_tile_blocks(A, v.shape[1], l.taps))
            it.nbias = l.cout
            it.db, it.gb = self.db.data_ptr() + odb * 4, self.gb.data_ptr() + ob * 4
            w1 = self.w1[ow:ow + n]
            w2 = self.w2[ow:ow + n]
            dw = self.dw[odw:odw + n]                      # copy 0; copies 1..R-1 follow at stride n
            if l.kind == 'conv':           # v (Cout, Cin, T): a = co, b = ci
                it.s1[0], it.s1[1], it.s1[2] = l.cout * l.cin, l.cin, 1          # [T][Cout][Cin]  forward + dW
                it.s2[0], it.s2[1], it.s2[2] = l.cout * l.cin, 1, l.cout          # [T][Cin][Cout]  data gradient
                l.wf, l.wb = w1.view(l.taps, l.cout, l.cin), w2.view(l.taps, l.cin, l.cout)
                l.dw = dw.view(l.taps, l.cout, l.cin)
            else:                          # v (Cin, Cout, T): a = ci, b = co
                it.s1[0], it.s1[1], it.s1[2] = l.cout * l.cin, l.cout, 1          # [T][Cin][Cout]  data gradient + dW
                it.s2[0], it.s2[1], it.s2[2] = l.cout * l.cin, 1, l.cin           # [T][Cout][Cin]  forward
                l.wb, l.wf = w1.view(l.taps, l.cin, l.cout), w2.view(l.taps, l.cout, l.cin)
                l.dw = dw.view(l.taps, l.cin, l.cout)
            l.db = self.db[odb:odb + l.cout]
            l.gb_view = self.gb[ob:ob + l.cout]
            l.gv_view = self.gv[ow:ow + n].view_as(v)
            l.gg_view = self.gg[oa:oa + A].view_as(g) if g is not None else None
            ow, oa, ob, blk = ow + pad8(n), oa + A, ob + l.cout, blk + A
            odw, odb = odw + pad8(n * R), odb + pad8(l.cout * R)
        self.total_blocks, self.total_tile_blocks = blk, tblk
        self.max_taps = max(l.taps for l in self.layers)
        self.max_row = max(l.weight.shape[1] * l.taps for l in self.layers)      # longest normalised row (parameters)
        raw = bytes(items)
        self._items_host = items
        self._copy_checks = 6              # backward passes after which idle privatised copies are looked for
        self.items_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.maps = _wn_maps(items, dev)
        self.dtype = dtype
        self._build_version = getattr(self, '_build_version', 0) + 1       # (prepare_together: combined tables are rebuilt)

    def _weight_params(self):
        for l in self.layers:
            yield l.weight
            if not l.plain:
                yield l.module.weight_g

    def _versions(self):
        return tuple(p._version for p in self._weight_params())

    def param_ptrs(self):
        if self._ptrs is None or self._ptrs[0] != self._sig:
            self._ptrs = (self._sig, frozenset(p.data_ptr() for p in self._weight_params()))
        return self._ptrs[1]

    def _stale(self, dtype, force=False):
        """(build the device buffers where parameters moved / the dtype changed, clear what a failed pass left behind and)
        tell whether the kernel-layout weight images must be refreshed now: (stale, versions, capturing)"""
        sig = self._signature(dtype)
        if sig != self._sig:
            self._build(dtype)
            self._sig = sig
            self.dirty = True
        if self._pending_w or self.deferred.n:
            # a backward pass that raised before its end-of-pass callback (no forward of a bank starts while its backward
            # runs): its waiting weight gradients and the recorded second stages of its partial sums must not ride along
            # with the next pass
            self._pending_w.clear()
            self.deferred.reset()
            self._queued = False
            self._touched = set()
            del self._hold[:]
            self._dw_stream.clear()
            self._open_nodes, self._side_used, self._finish_stream = 0, [], None
            self._close_task, self._close_mixed = None, False
        versions = self._versions()
        capturing = self.w1.is_cuda and torch.cuda.is_current_stream_capturing()
        stale = self.eager_stale and not capturing
        clean = SKIP_CLEAN_PREPARE and not force and not self.dirty and not stale and versions == self._clean_versions
        return (not clean), versions, capturing

    def _refreshed(self, versions, capturing):
        self.dirty, self._clean_versions = False, versions
        if not capturing:
            self.eager_stale = False

    def prepare(self, dtype, force=False):
        """Refresh kernel-layout weights from (weight_v, weight_g): one call (two launches) for the whole network -- skipped
        while the bank is clean (see SKIP_CLEAN_PREPARE)."""
        stale, versions, capturing = self._stale(dtype, force)
        if not stale:
            return
        _wn_prepare(self.items_dev, self.maps, len(self.layers), self.total_blocks, self.total_tile_blocks, self.max_taps,
                    lib.stream(self.w1), 'msmc_wn_prepare_multi_tiles')
        self._refreshed(versions, capturing)

    # -- end-of-backward: kernel-layout dW -> parameter gradients --------------------------------------
    def _queue_finish(self):
        if not self._queued:
            self._queued = True
            Variable._execution_engine.queue_callback(self._finish_backward)

    @staticmethod
    def _grad_pairs(l):
        m = l.module
        return (((m.bias, l.gb_view), (m.weight, l.gv_view)) if l.plain else
                ((m.bias, l.gb_view), (m.weight_g, l.gg_view), (m.weight_v, l.gv_view)))

    _NO_ATOMICS = frozenset((3, 4, 5, 6, 7, 9))       # msmc_conv_wgrad variants that accumulate through a second stage into copy 0

    def _drop_idle_copies(self):
        """Privatised dW / db copies (DW_COPIES per small layer) exist for the ATOMIC weight-gradient generations: same-address
        atomics retire serially, copies shorten the chain.  A layer whose weight gradients only ever ran on the no-atomics
        generations writes copy 0 alone, yet the weight-norm backward read, summed and re-zeroed all eight -- more than half of
        that pass's traffic.  After the tuner has settled (a few passes), such layers go down to one copy: the device item table
        is patched in place (same address: captured graphs keep working).  A later shape that does pick an atomic kernel then
        simply runs with one copy."""
        if self._copy_checks <= 0 or (self.w1.is_cuda and torch.cuda.is_current_stream_capturing()):
            return
        self._copy_checks -= 1
        changed = False
        for l, it in zip(self.layers, self._items_host):
            if l.dw_copies > 1 and l.wg_variants and l.wg_variants <= self._NO_ATOMICS:
                l.dw_copies = 1
                it.copies = 1
                changed = True
        if changed:
            self.items_dev.copy_(torch.frombuffer(bytearray(bytes(self._items_host)), dtype=torch.uint8))

    def node_opened(self):
        self._open_nodes += 1

    def node_closed(self):
        """a backward node with a weight gradient has issued it; the last one of the pass completes the bank.
        The count is one number per bank, not one per backward pass: when nodes recorded by SEVERAL forward passes of the bank
        are closed by different backward passes (two graphs over one bank, back-propagated one after the other), a count that
        reaches zero says nothing about the pass that is running -- such a bank is recognised by the autograd graph-task id
        of its closing nodes and delivers in the end-of-pass callback, as every bank did before round 4."""
        task = torch._C._current_graph_task_id() if hasattr(torch._C, '_current_graph_task_id') else -1
        if self._open_nodes > 0:
            if self._close_task is None:
                self._close_task = task
            elif self._close_task != task:
                self._close_mixed = True
            self._open_nodes -= 1
            if self._open_nodes == 0 and self._close_mixed:
                self._close_task, self._close_mixed = None, False
                return
            # (a bank whose backward nodes run on several streams finishes in the end-of-pass callback, on the caller's stream:
            # nothing orders the caller behind whichever branch happened to close last)
            if self._open_nodes == 0 and EARLY_FINISH and self._touched and not self.streams:
                # The bank's second stages and its weight-norm backward have no reader before the optimizer: on a GPU they
                # leave on a side stream of the library's own (ordered behind everything the calling stream has issued), so
                # that the backward chain of the networks upstream -- the frame decoder, quantiser and encoders behind the
                # vocoder -- does not queue behind them; the end-of-pass callback joins that stream.
                dev = self.w1.device
                if FINISH_SIDE and dev.type == 'cuda' and STREAMS_ENABLED:
                    fin = own_streams(dev, 1, 'finish')[0]
                    fin.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(fin):
                        self._finish_backward(early=True)
                    self._finish_stream = fin
                else:
                    self._finish_backward(early=True)

    def _finish_backward(self, early=False):
        self._close_task, self._close_mixed = None, False      # (early: the count is back at zero; otherwise the pass is over)
        if not early:
            self._queued = False
            self._open_nodes = 0        # (whatever the count missed -- unused outputs -- ends with the pass)
            if self._finish_stream is not None:       # the early delivery of this pass ran on a side stream: join it
                torch.cuda.current_stream(self.w1.device).wait_stream(self._finish_stream)
                self._finish_stream = None
        self.flush_wgrad()
        if self._side_used:             # weight-gradient branches join here, before their inputs are released
            cur = torch.cuda.current_stream(self.w1.device)
            for st in self._side_used:
                cur.wait_stream(st)
            self._side_used = []
        self._dw_stream.clear()
        if not early:                   # (held gradients may still be read by other banks' nodes: released with the pass)
            del self._hold[:]
        touched, self._touched = self._touched, set()
        if self.streams and self.w1.is_cuda:      # backward launches ran on the side streams of their forward
            cur = torch.cuda.current_stream(self.w1.device)
            for st in self.streams:
                cur.wait_stream(st)
        if not touched:                 # a pass that only propagated through this network (frozen D in the G step),
            self.deferred.flush(lib.stream(self.w1))       # or one whose gradients were delivered early
            if not early:
                self._drop_idle_copies()
            return
        # the second stage of every no-atomics weight gradient of this pass, merged (the partial results sat in the arena)
        self.deferred.flush(lib.stream(self.w1))
        with torch.no_grad():
            # torch .grad semantics: a gradient that is still live (no zero_grad since the last backward) is added to.
            # The live gradients ARE the bank's output buffers, so the kernel accumulates in place; buffers of
            # parameters whose gradient was reset are cleared first.
            live = lambda p, gview: p.grad is not None and p.grad.data_ptr() == gview.data_ptr()
            accumulate = any(live(p, gv) for l in self.layers for p, gv in self._grad_pairs(l))
            if accumulate:
                for l in self.layers:
                    for p, gview in self._grad_pairs(l):
                        if not live(p, gview):
                            gview.zero_()
            lib.check(lib.get().msmc_wn_backward_multi_rows(lib.ptr(self.items_dev), len(self.layers), self.total_blocks,
                                                            1 if accumulate else 0, self.max_row, lib.stream(self.w1)),
                      'msmc_wn_backward_multi_rows')
            for l in self.layers:
                if l.index not in touched or not l.weight.requires_grad:
                    continue
                # gradients ARE the bank's output buffers (no copies): they stay valid until the next backward
                # of this network, i.e. past the optimizer step that consumes them
                for p, gview in self._grad_pairs(l):
                    if p.grad is None:
                        p.grad = gview
                    elif p.grad.data_ptr() != gview.data_ptr():
                        p.grad.add_(gview)
                    if GRAD_READY_HOOK is not None:
                        GRAD_READY_HOOK(p)
            if not early:
                self._drop_idle_copies()


def _tap_grad(g_tap, like):
    """gradient that arrived through a tap (an alias of a convolution's input handed to another consumer): contiguous,
    in the dtype of the data gradient it is added to"""
    if g_tap is None:
        return None
    if g_tap.dtype != like.dtype:
        g_tap = g_tap.to(like.dtype)
    return g_tap.contiguous()


class _HipConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, res2, weight_token, bank, layer, in_slope, out_slope, out_div, tap=False, in_act=1.0,
                out_masked=False, defer_out=False, in_grad_div=1.0, grad_predivided=False):
        # ``out_div`` in the backward pass is a division of the whole incoming gradient -- a stock tensor pass per use.  Where
        # the output has ONE consumer and that consumer is a convolution, the consumer's data-gradient launch delivers the
        # gradient already divided (its ``in_grad_div`` = this node's ``out_div``: the division runs in that launch's epilogue)
        # and this node is told so (``grad_predivided``).
        # ``in_act`` != 1: x was activated by its producer's epilogue (that producer ran with out_slope = in_act and
        # ``out_masked``); forward and weight gradient read it as it is, the data gradient applies the activation's
        # derivative (sign of the activated value = sign of the pre-activation).  ``out_masked``: this convolution's ONLY
        # consumer does that, so the backward pass here does not.  With ``tap`` on such an input (reflect-padded layers
        # only: the resolution discriminators' feature maps) the tap's gradient is that of another reader of the ACTIVATED
        # map and goes through the derivative as well: (fold + tap) * lrelu'.
        m = layer.module
        assert in_act == 1.0 or (in_slope == 1.0 and (layer.reflect or not tap))
        if defer_out:
            # the forward product is computed by the CONSUMER's launch (hip_conv_add_ln: the fused projection + add + LayerNorm
            # kernel reads x and the weight slice itself): this node only holds the layer's place in the autograd graph --
            # its backward pass (data gradient, weight gradient, the bank's bookkeeping) is the ordinary one
            assert layer.kind == 'conv' and layer.taps == 1 and res is None and res2 is None and not tap
            assert in_slope == 1.0 and out_slope == 1.0 and out_div == 1.0 and in_act == 1.0
            out = x.new_empty(x.shape[:-1] + (layer.cout,))
        elif layer.kind == 'conv':
            geom = layer.geom(x.shape[1], x.shape[2])
            out = K.conv_forward(x, layer.wf, geom, bias=m.bias, in_slope=in_slope, res=res, res2=res2,
                                 out_div=out_div, out_slope=out_slope)
        else:
            assert res is None and res2 is None and out_div == 1.0 and out_slope == 1.0
            out = K.conv_transpose1d_forward(x, layer.wf, layer.kernel[1], layer.stride[1], layer.padding[1],
                                             bias=m.bias, in_slope=in_slope)
        ctx.bank, ctx.layer = bank, layer
        ctx.in_slope, ctx.out_slope, ctx.out_div = in_slope, out_slope, out_div
        ctx.mask_slope = in_act if in_act != 1.0 else in_slope
        ctx.out_masked = bool(out_masked)
        ctx.in_grad_div, ctx.grad_predivided = float(in_grad_div), bool(grad_predivided)
        assert ctx.in_grad_div == 1.0 or not (tap or layer.reflect)
        ctx.has_res, ctx.has_res2 = res is not None, res2 is not None
        ctx.need_w = layer.weight.requires_grad
        ctx.counted = bool(ctx.need_w and ctx.needs_input_grad[3])      # (a graph is being recorded and the weight is in it)
        if ctx.counted:
            bank.node_opened()
        ctx.save_for_backward(x, out if (out_slope != 1.0 and not out_masked) else None)
        ctx.set_materialize_grads(False)
        # ``tap``: also return an alias of x for x's OTHER consumer (a residual add, a feature-matching loss, a fused
        # LayerNorm's residual input).  Its gradient then arrives HERE and is added in the data-gradient launch's epilogue
        # instead of by a stock add_ of the autograd engine (one launch and one read-modify-write of the tensor less).
        return (out, x.view_as(x)) if tap else out

    @staticmethod
    def backward(ctx, g, g_tap=None):
        x, out = ctx.saved_tensors
        layer, bank = ctx.layer, ctx.bank
        if g is None:                                  # only the tap was used
            if g_tap is not None and ctx.mask_slope != ctx.in_slope:
                # the tap's gradient is that of a reader of the ACTIVATED map (the producer ran with out_masked and leaves
                # the derivative to this node): the grouped form refuses the case, so does this one
                raise RuntimeError('hip_conv: output unused but its tap carries a gradient through an activated input')
            if ctx.counted:
                ctx.counted = False
                bank.node_closed()
            return (g_tap,) + (None,) * 14
        g = g.contiguous()
        g_tap = _tap_grad(g_tap, g)
        if ctx.out_slope != 1.0 and not ctx.out_masked:        # y = lrelu(z): dz = dy * (y > 0 ? 1 : slope)
            g = K.lrelu_bwd(g, out, ctx.out_slope)
        if ctx.out_div != 1.0 and not ctx.grad_predivided:
            g = g / ctx.out_div
        gx = None
        if ctx.needs_input_grad[0]:
            mask = x if ctx.mask_slope != 1.0 else None
            if layer.kind == 'conv':
                geom = layer.geom(x.shape[1], x.shape[2])
                if layer.reflect and ctx.mask_slope != ctx.in_slope:         # activated input: (fold + tap) * lrelu'
                    gp = K.conv_dgrad(g, layer.wb, geom)
                    gx = K.reflect_fold_group([(gp, x.shape[1], x.shape[2], mask, g_tap)], layer.padding[0],
                                              ctx.mask_slope, tap_first=True)[0]
                elif layer.reflect:
                    gp = K.conv_dgrad(g, layer.wb, geom)
                    gx = K.reflect_fold(gp, x.shape[1], x.shape[2], layer.padding[0], mask_src=mask,
                                        slope=ctx.in_slope)
                    if g_tap is not None:
                        gx = gx + g_tap
                else:
                    gx = K.conv_dgrad(g, layer.wb, geom, mask_src=mask, mask_slope=ctx.mask_slope, res=g_tap,
                                      out_div=ctx.in_grad_div)
            else:
                gx = K.conv_transpose1d_dgrad(g, layer.wb, layer.kernel[1], layer.stride[1], layer.padding[1],
                                              x.shape[2], mask_src=mask, mask_slope=ctx.mask_slope, res=g_tap,
                                              out_div=ctx.in_grad_div)
            if g_tap is not None:
                bank._hold.append(g_tap)               # (read by a launch that may replay on another stream)
                bank._queue_finish()
        elif g_tap is not None:
            gx = g_tap
        if ctx.need_w:
            if layer.kind == 'conv' and WGRAD_BATCH > 1:       # waits in the bank for company (see WGRAD_BATCH)
                bank.queue_wgrad(dict(x=x, g=g, geom=layer.geom(x.shape[1], x.shape[2]), n_slices=layer.taps,
                                      in_slope=ctx.in_slope, dw=layer.dw, db=layer.db, copies=layer.dw_copies, seen=layer.wg_variants))
            else:
                with bank.wgrad_side(x, g, dws=(layer.dw,)):
                    if layer.kind == 'conv':
                        K.conv_wgrad(x, g, layer.geom(x.shape[1], x.shape[2]), layer.taps, in_slope=ctx.in_slope,
                                     dw=layer.dw, db=layer.db, copies=layer.dw_copies, seen=layer.wg_variants)
                    else:
                        K.conv_transpose1d_wgrad(x, g, layer.kernel[1], layer.stride[1], layer.padding[1],
                                                 in_slope=ctx.in_slope, dw=layer.dw, copies=layer.dw_copies, seen=layer.wg_variants)
                        K.colsum(g.reshape(-1, g.shape[-1]), out=layer.db)
            bank._touched.add(layer.index)
            bank._queue_finish()
        if ctx.has_res or ctx.has_res2:
            # g goes to several consumers that may replay on different streams.  The autograd engine accumulates
            # IN PLACE into a gradient it holds the only reference to, while other streams may still be reading
            # that same tensor (their out-of-place sums are ordered only against the producer).  Keeping a
            # reference until the end of the backward pass rules the in-place path out for this tensor.
            bank._hold.append(g)
            bank._queue_finish()
        if ctx.counted:
            ctx.counted = False
            bank.node_closed()
        # weight_token (the layer's weight_v) only ties the output to the parameters in the autograd graph;
        # parameter gradients are produced in kernel layout and delivered by ConvBank._finish_backward.
        return (gx, (g if ctx.has_res else None), (g if ctx.has_res2 else None)) + (None,) * 12


class _HipConvGroup(torch.autograd.Function):
    """Several independent convolutions of one bank as grouped launches (K.conv_forward_group / conv_dgrad_group /
    conv_wgrad_group): the parallel ResBlocks of a generator stage, one layer of all period / resolution
    sub-discriminators.  ``specs[k] = (layer, in_slope, out_slope, out_div, has_res, has_res2, tap, in_act, out_masked)``
    (the last two as in _HipConv.forward); ``tensors`` is the flattened list x_k, [res_k], [res2_k], weight_token_k."""

    @staticmethod
    def forward(ctx, bank, specs, *tensors):
        items, pos, members = [], 0, []
        for layer, in_slope, out_slope, out_div, has_res, has_res2, tap, in_act, out_masked in specs:
            assert in_act == 1.0 or (in_slope == 1.0 and (layer.reflect or not tap))
            x = tensors[pos]
            res = tensors[pos + 1] if has_res else None
            res2 = tensors[pos + 1 + has_res] if has_res2 else None
            members.append((pos, x))
            pos += 2 + has_res + has_res2
            assert layer.kind == 'conv'
            items.append(dict(x=x, w=layer.wf, geom=layer.geom(x.shape[1], x.shape[2]), bias=layer.module.bias,
                              in_slope=in_slope, res=res, res2=res2, out_div=out_div, out_slope=out_slope))
        outs = K.conv_forward_group(items)
        ctx.bank, ctx.specs, ctx.ntensors = bank, specs, len(tensors)
        ctx.need_w = [sp[0].weight.requires_grad for sp in specs]     # as of the forward (a frozen pass stays frozen)
        ctx.xpos = [m[0] for m in members]
        # (weight token of member k: the last of its tensors) -- counted when a graph is being recorded with a weight in it
        wpos = [m[0] + 1 + sp[4] + sp[5] for m, sp in zip(members, specs)]
        ctx.counted = any(nw and ctx.needs_input_grad[2 + wp] for nw, wp in zip(ctx.need_w, wpos))
        if ctx.counted:
            bank.node_opened()
        saved = [m[1] for m in members] + [o if (sp[2] != 1.0 and not sp[8]) else None for o, sp in zip(outs, specs)]
        ctx.save_for_backward(*saved)
        ctx.set_materialize_grads(False)
        # members with ``tap``: an alias of their input follows the outputs (see _HipConv.forward)
        ctx.tapped = [k for k, sp in enumerate(specs) if sp[6]]
        return tuple(outs) + tuple(members[k][1].view_as(members[k][1]) for k in ctx.tapped)

    @staticmethod
    def backward(ctx, *gs):
        n = len(ctx.specs)
        saved = ctx.saved_tensors
        xs, outs = saved[:n], saved[n:]
        bank = ctx.bank
        grads = [None] * ctx.ntensors
        gl, d_items, d_members, w_items = [], [], [], []
        g_taps = dict(zip(ctx.tapped, gs[n:]))
        gs = list(gs[:n])
        for k in range(n):
            if gs[k] is None:                          # an output nobody used: zero gradient
                gs[k] = torch.zeros_like(outs[k]) if outs[k] is not None else None
        if any(g is None for g in gs):
            raise RuntimeError('grouped convolution: an output without gradient (not expected on the training path)')
        gs = [g.contiguous() for g in gs]
        # y = lrelu(z): dz = dy * (y > 0 ? 1 : slope) -- one multi-tensor launch per slope value
        for slope in sorted(set(sp[2] for sp in ctx.specs if sp[2] != 1.0 and not sp[8])):
            ks = [k for k, sp in enumerate(ctx.specs) if sp[2] == slope and not sp[8]]
            for k, gm in zip(ks, K.lrelu_bwd_group([(gs[k], outs[k]) for k in ks], slope)):
                gs[k] = gm
        for k, (layer, in_slope, out_slope, out_div, has_res, has_res2, tap, in_act, out_masked) in enumerate(ctx.specs):
            mask_slope = in_act if in_act != 1.0 else in_slope
            g = gs[k]
            if out_div != 1.0:
                g = g / out_div
            gl.append(g)
            x = xs[k]
            geom = layer.geom(x.shape[1], x.shape[2])
            pos = ctx.xpos[k]
            if ctx.needs_input_grad[2 + pos]:
                mask = x if mask_slope != 1.0 else None
                g_tap = _tap_grad(g_taps.get(k), g)
                if layer.reflect:
                    d_items.append(dict(g=g, wb=layer.wb, geom=geom))
                else:
                    d_items.append(dict(g=g, wb=layer.wb, geom=geom, mask_src=mask, mask_slope=mask_slope, res=g_tap))
                    if g_tap is not None:
                        bank._hold.append(g_tap)
                d_members.append(k)
            if ctx.need_w[k]:
                w_items.append(dict(x=x, g=g, geom=geom, n_slices=layer.taps, in_slope=in_slope, dw=layer.dw, db=layer.db,
                                    copies=layer.dw_copies, seen=layer.wg_variants))
                bank._touched.add(layer.index)
            if has_res:
                grads[pos + 1] = g
            if has_res2:
                grads[pos + 1 + has_res] = g
            if has_res or has_res2:
                bank._hold.append(g)                  # shared gradient: see _HipConv.backward
        if d_items:
            folds = {}
            for k, gx in zip(d_members, K.conv_dgrad_group(d_items)):
                layer, in_slope = ctx.specs[k][0], ctx.specs[k][1]
                if layer.reflect:           # gradient on the padded grid: fold the border back (multi-tensor launch)
                    x = xs[k]
                    g_tap = _tap_grad(g_taps.get(k), gx)
                    if g_tap is not None:
                        bank._hold.append(g_tap)
                    in_act = ctx.specs[k][7]
                    slope = in_act if in_act != 1.0 else in_slope
                    folds.setdefault((layer.padding[0], slope, in_act != 1.0), []).append(
                        (k, (gx, x.shape[1], x.shape[2], x if slope != 1.0 else None, g_tap)))
                else:
                    grads[ctx.xpos[k]] = gx
            for (pad, slope, tap_first), members in folds.items():
                for (k, _), gx in zip(members, K.reflect_fold_group([m[1] for m in members], pad, slope,
                                                                     tap_first=tap_first)):
                    grads[ctx.xpos[k]] = gx
        if w_items and WGRAD_BATCH_GROUPS and WGRAD_BATCH > 1:
            for it in w_items:
                bank.queue_wgrad(it)
        elif w_items:
            with bank.wgrad_side(*([it['x'] for it in w_items] + [it['g'] for it in w_items]), dws=[it['dw'] for it in w_items]):
                K.conv_wgrad_group(w_items)
        # members that read the SAME tensor (the parallel ResBlocks of a generator stage all start from the stage's input): their
        # input gradients leave as one sum -- one launch -- instead of as separate gradients the autograd engine adds pairwise
        if SUM_SHARED_INPUTS:
            shared = {}
            for k in range(n):
                if grads[ctx.xpos[k]] is not None:
                    # the same autograd tensor, not merely the same memory: a tap alias of x shares x's storage but its gradient
                    # takes another route (through its producer's data-gradient epilogue, possibly an activation's derivative)
                    key = (xs[k].data_ptr(), tuple(xs[k].shape), id(xs[k].grad_fn), xs[k].output_nr, xs[k].is_leaf)
                    shared.setdefault(key, []).append(ctx.xpos[k])
            for slots in shared.values():
                if 2 <= len(slots) <= 4:
                    from . import norm
                    parts = [grads[p] for p in slots]
                    grads[slots[0]] = norm.sum_n(parts)
                    for p in slots[1:]:
                        grads[p] = None
                    bank._hold.extend(parts)           # (read by a launch that may replay on another stream)
        bank._queue_finish()
        if ctx.counted:
            ctx.counted = False
            bank.node_closed()
        return (None, None) + tuple(grads)


def hip_conv_group(bank, members):
    """``members``: list of dicts(layer=, x=, res=None, res2=None, in_slope=1, out_slope=1, out_div=1) -- independent
    convolutions issued together.  Returns the outputs in order."""
    specs, tensors = [], []
    for m in members:
        res, res2 = m.get('res'), m.get('res2')
        specs.append((m['layer'], float(m.get('in_slope', 1.0)), float(m.get('out_slope', 1.0)),
                      float(m.get('out_div', 1.0)), int(res is not None), int(res2 is not None), bool(m.get('tap', False)),
                      float(m.get('in_act', 1.0)), bool(m.get('out_masked', False))))
        tensors.append(m['x'])
        if res is not None:
            tensors.append(res)
        if res2 is not None:
            tensors.append(res2)
        tensors.append(m['layer'].weight)
    outs = list(_HipConvGroup.apply(bank, tuple(specs), *tensors))
    n = len(members)
    if len(outs) == n:
        return outs
    taps = iter(outs[n:])                      # members with tap=True return (out, tap) instead of out
    return [(outs[k], next(taps)) if specs[k][6] else outs[k] for k in range(n)]


# grouped launches replace the fork/join streams (forked hipGraph branches do not overlap; one grid does)
GROUPED = os.environ.get('MSMC_GROUPED', '1') != '0'


def hip_conv(bank, layer, x, res=None, res2=None, in_slope=1.0, out_slope=1.0, out_div=1.0, tap=False, in_act=1.0,
             out_masked=False, in_grad_div=1.0, grad_predivided=False):
    """``tap=True``: returns (out, x_tap) -- x_tap aliases x and is what x's other consumer should read, so that its
    gradient is added inside this convolution's data-gradient launch (see _HipConv.forward).
    Activation in the PRODUCER's epilogue: ``a = hip_conv(.., out_slope=s, out_masked=True)`` followed by
    ``hip_conv(.., a, in_act=s)`` computes conv(lrelu_s(conv(..))) with the activation applied once, where the value is
    produced, instead of in every load of the consumer (a must have no other consumer).
    ``y = hip_conv(.., out_div=n, grad_predivided=True)`` followed by ``hip_conv(.., y, in_grad_div=n)`` (y's only
    consumer): the backward division by n runs in the consumer's data-gradient epilogue (see _HipConv.forward)."""
    return _HipConv.apply(x, res, res2, layer.weight, bank, layer, float(in_slope), float(out_slope),
                          float(out_div), bool(tap), float(in_act), bool(out_masked), False, float(in_grad_div),
                          bool(grad_predivided))


# 1 (default): see _HipConvGroup.backward (input gradients of members that share their input tensor)
SUM_SHARED_INPUTS = os.environ.get('MSMC_SUM_SHARED_INPUTS', '1') != '0'

# 1 (default): see _HipConv.forward ``in_grad_div`` (the generator's mean over its parallel ResBlocks)
GRAD_DIV_FUSE = os.environ.get('MSMC_GRAD_DIV_FUSE', '1') != '0'

# 1 (default): a 1-tap projection whose only consumer is a fused add + LayerNorm runs INSIDE that launch (csrc/norm.hip
# fc_add_ln_fwd_kernel: the attention sub-layer's output projection, 12 launches and 12 round trips of h per forward pass
# less); MSMC_FC_LN_FUSE=0 keeps the two launches (A/B, and what fp32 and shapes the kernel does not take always run)
FC_LN_FUSE = os.environ.get('MSMC_FC_LN_FUSE', '1') != '0'


def fc_ln_fusable(layer, x):
    return (FC_LN_FUSE and x.dtype == torch.bfloat16 and layer.kind == 'conv' and layer.taps == 1 and x.shape[1] == 1 and
            layer.cin % 32 == 0 and layer.cout % 4 == 0 and layer.cout <= 640 and layer.module.bias is not None)


def hip_conv_add_ln(bank, layer, x, res, gamma, beta, keep_row=None, p_drop=0.0, salt=0, eps=1e-5):
    """layer_norm(dropout(conv_1tap(x) + bias) + res) * keep_row for x [B, 1, T, Cin], res [B, T, Cout]: one launch where the
    fused kernel takes the case, hip_conv followed by hip/norm.py add_layer_norm otherwise (same masks, same saved tensors,
    same backward nodes either way)"""
    from . import norm
    aligned = all(t is not None and t.data_ptr() % 16 == 0 for t in (layer.module.bias, gamma, beta))
    if not (fc_ln_fusable(layer, x) and aligned):
        return norm.add_layer_norm(hip_conv(bank, layer, x).squeeze(1), res, gamma, beta, keep_row=keep_row, p_drop=p_drop,
                                   salt=salt, eps=eps)
    h = _HipConv.apply(x, None, None, layer.weight, bank, layer, 1.0, 1.0, 1.0, False, 1.0, False, True).squeeze(1)
    return norm.add_layer_norm(h, res, gamma, beta, keep_row=keep_row, p_drop=p_drop, salt=salt, eps=eps,
                               fc=(x, layer.wf, layer.module.bias))
