"""Data parallelism over RCCL/xGMI (re-expression of reference msmctts/distributed/distributed.py:21-31,154-204).

The reference broadcasts every ``state_dict`` tensor separately at start-up (636 messages) and, after
each *whole* backward, flattens all gradients into one buffer, all-reduces it, divides and copies it
back.  Here, for one process per GPU on a fully connected xGMI node:

* ``init_distributed`` keeps the reference signature; backend "nccl" is RCCL on ROCm.
* start-up sync is ONE flat broadcast per dtype (params + buffers, VQ codebooks included);
* gradients are grouped into size-bounded buckets per top-level child (``autoencoder`` /
  ``discriminator`` never share a bucket, so the D step and the G step each complete their own
  buckets); a bucket is all-reduced asynchronously on RCCL's stream as soon as its last gradient has
  been accumulated, i.e. overlapped with the rest of backward; ``GradReducer.finish()`` (called by the
  trainer after ``backward()``) flushes partially filled buckets, waits, and writes the averaged
  values back.  Parameters that received no gradient (frozen tables, the vocoder during warm-up, the
  discriminator during the G step) are simply not communicated -- the reference reduces the stale D
  gradients on every G backward.
* hipGraph mode has two exchanges: ``serial`` (default) -- the hooks are off and ONE flat all-reduce per child runs between
  the replayed segments (``allreduce_child``) -- and ``overlap`` (``VQGANTrainer.graph_exchange`` / MSMC_GRAPH_EXCHANGE=overlap)
  -- the hooks stay on DURING CAPTURE, so the bucketed all-reduces are recorded into the segment's graph on RCCL's stream as a
  forked branch and ``finish()`` is recorded at the segment's end; together with the banks' early gradient delivery
  (hip/convnet.py ``ConvBank._open_nodes``) the generator's buckets travel under the frame decoder's / quantiser's /
  encoders' backward.  ``overlap`` has only ever run on ONE GPU (world-1 RCCL capture, tests/test_gpu_parity.py): no
  multi-GPU node was reachable, which is why it is not the default.
* VQ codebook statistics are, like the reference, NOT synchronised by default (each rank EMA-updates
  from its local batch; rank 0's codebook is checkpointed).
"""
import os

import torch
import torch.distributed as dist

DEFAULT_BUCKET_BYTES = 32 * 1024 * 1024
# Wire format of the gradient exchange: fp32 (the reference's, default) or bf16 (MSMC_GRAD_EXCHANGE=bf16 / the
# ``exchange_dtype`` argument): half the bytes on xGMI for ~3 significant digits per summand -- the sum is formed by
# RCCL in bf16, the division by the world size and everything after it in fp32.  Budget per step at CSMSC sizes
# (DESIGN.md section 6): 51 MB (D) + 147 MB (autoencoder) in fp32.
_EXCHANGE = {'fp32': torch.float32, 'float32': torch.float32, 'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16}
DEFAULT_EXCHANGE_DTYPE = _EXCHANGE[os.environ.get('MSMC_GRAD_EXCHANGE', 'fp32')]


def init_distributed(rank, num_gpus, group_name, dist_backend, dist_url):
    """One process per GPU; same arguments as the reference (group_name is accepted and unused)."""
    if dist_backend == 'nccl':
        assert torch.cuda.is_available(), 'Distributed mode requires a GPU.'
        torch.cuda.set_device(rank % max(1, torch.cuda.device_count()))
    if not dist.is_initialized():
        dist.init_process_group(dist_backend, init_method=dist_url, world_size=num_gpus, rank=rank)


def broadcast_state(module, src=0):
    """Coalesced start-up broadcast of every tensor in ``state_dict`` (one message per dtype)."""
    by_dtype = {}
    for t in module.state_dict().values():
        if torch.is_tensor(t):
            by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for tensors in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.broadcast(flat, src)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


class _Bucket(object):
    def __init__(self, params):
        self.params = params
        self.ready = []
        self.pending = set(id(p) for p in params)
        self.streams = []            # streams the bucket's gradients became ready on (see GradReducer._on_grad)


class GradReducer(object):
    """Bucketed, backward-overlapped gradient averaging."""

    def __init__(self, module, bucket_bytes=DEFAULT_BUCKET_BYTES, group=None, exchange_dtype=None):
        self.group = group
        self.exchange_dtype = exchange_dtype or DEFAULT_EXCHANGE_DTYPE
        self.world = dist.get_world_size(group)
        self.buckets = []
        self.hooks_enabled = True
        self._owner = {}
        self._inflight = []
        self._flat = {}                      # per child: persistent flat exchange buffer of the graph-mode path
        children = list(module.named_children()) or [('', module)]
        for _, child in children:
            params = [p for p in child.parameters() if p.requires_grad]
            cur, size = [], 0
            for p in reversed(params):               # gradients arrive roughly in reverse registration order
                cur.append(p)
                size += p.numel() * p.element_size()
                if size >= bucket_bytes:
                    self._add(cur)
                    cur, size = [], 0
            if cur:
                self._add(cur)
        for b in self.buckets:
            for p in b.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def _add(self, params):
        b = _Bucket(list(params))
        self.buckets.append(b)
        for p in params:
            self._owner[id(p)] = b

    def _on_grad(self, p):
        if not self.hooks_enabled:           # hipGraph mode: gradients are exchanged between the replayed segments
            return
        b = self._owner.get(id(p))
        if b is None or p.grad is None:      # e.g. the token edge of a HIP conv: its gradient arrives by hand later
            return
        if id(p) in b.pending:
            b.pending.discard(id(p))
            b.ready.append(p)
            # A gradient is ready ON THE STREAM ITS HOOK FIRES ON, not on whichever stream later completes the bucket: a
            # convolution bank delivers its gradients from a side stream of its own (hip/convnet.py ConvBank.node_closed,
            # FINISH_SIDE) while the parameters of stock modules in the same bucket become ready on the calling stream --
            # the stream that launches the bucket waits for every stream that contributed to it.
            if p.grad.is_cuda:
                st = torch.cuda.current_stream(p.grad.device)
                if not any(st == u for u in b.streams):
                    b.streams.append(st)
            if not b.pending:
                self._launch(b)

    def _launch(self, b):
        ps = [p for p in b.params if any(p is r for r in b.ready)]   # fixed (registration) order on every rank
        if ps:
            if b.streams:
                cur = torch.cuda.current_stream(ps[0].grad.device)
                for st in b.streams:
                    if st != cur:
                        cur.wait_stream(st)
            flat = torch.cat([p.grad.reshape(-1) for p in ps]).to(self.exchange_dtype)
            work = dist.all_reduce(flat, group=self.group, async_op=True)
            self._inflight.append((work, flat, ps))
        b.ready = []
        b.streams = []
        b.pending = set(id(p) for p in b.params)

    def allreduce_child(self, child, grads=None):
        """Synchronous exchange of one child's gradients (used between hipGraph replays, where the hooks do not fire):
        one flat all-reduce, averaged, written back in place.  ``grads``: the static gradient tensors the graphs write
        (the trainer records them at capture, so that a loop resetting ``p.grad`` cannot make this a silent no-op);
        without it, every existing ``p.grad`` of the child."""
        if grads is None:
            grads = [p.grad for p in child.parameters() if p.requires_grad and p.grad is not None]
        if not grads:
            raise RuntimeError('gradient exchange found no gradients: ranks would silently diverge')
        flat = self._flat.get(id(child))
        total = sum(g.numel() for g in grads)
        if flat is None or flat.numel() != total or flat.device != grads[0].device or flat.dtype != self.exchange_dtype:
            flat = self._flat[id(child)] = torch.empty(total, dtype=self.exchange_dtype, device=grads[0].device)
        views = self._views(flat, grads)
        torch._foreach_copy_(views, grads)
        dist.all_reduce(flat, group=self.group)
        torch._foreach_copy_(grads, views)
        torch._foreach_div_(grads, float(self.world))           # (in the gradients' own fp32, whatever the wire format)
        for b in self.buckets:                      # drop hook state recorded while capturing
            b.ready = []
            b.streams = []
            b.pending = set(id(q) for q in b.params)
        self._inflight = []

    @staticmethod
    def _views(flat, grads):
        out, off = [], 0
        for g in grads:
            n = g.numel()
            out.append(flat[off:off + n].view_as(g))
            off += n
        return out

    def finish(self):
        """Flush partial buckets, wait for the collectives, write back grad / world_size."""
        for b in self.buckets:
            if b.ready:
                self._launch(b)
        for work, flat, ps in self._inflight:
            work.wait()             # (the calling stream waits for RCCL's: also what a hipGraph capture records)
            grads = [p.grad for p in ps]
            torch._foreach_copy_(grads, self._views(flat, grads))        # two multi-tensor launches per bucket, not two
            torch._foreach_div_(grads, float(self.world))                # per parameter (636 parameters per step)
        self._inflight = []


def apply_gradient_allreduce(module, bucket_bytes=DEFAULT_BUCKET_BYTES, exchange_dtype=None):
    """Reference entry point: sync initial state from rank 0 and arm gradient averaging.

    Returns the same module (no wrapper class, like the reference) with ``module.grad_reducer`` set;
    trainers call ``module.grad_reducer.finish()`` after each ``backward()``.
    """
    with torch.no_grad():
        broadcast_state(module, 0)
    module.grad_reducer = GradReducer(module, bucket_bytes, exchange_dtype=exchange_dtype)
    from ..hip import convnet
    convnet.GRAD_READY_HOOK = module.grad_reducer._on_grad      # HIP conv stacks deliver their gradients by hand
    return module
