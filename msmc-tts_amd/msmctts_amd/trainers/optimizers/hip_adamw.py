"""AdamW (+ gradient-norm clipping) over csrc/optim.hip: three launches per step for all tensors of a child
(reference trainers/optimizers/__init__.py:53-78 builds ``torch.optim.AdamW`` per child; msmctts_trainer.py:205-206 clips
the autoencoder first).  Same arithmetic and the same ``state_dict`` layout as ``torch.optim.AdamW`` (per-parameter
``step`` / ``exp_avg`` / ``exp_avg_sq``), so checkpoints move both ways.  Learning rate, step count and clip
coefficient live on the device: the step replays from a hipGraph unchanged.

Deviation from ``torch.optim.AdamW``: the step count is ONE device word per group, not one per parameter.  A parameter
that first receives a gradient at step k > 1 (none of the shipped models has one: every trainable tensor of a child gets a
gradient on every backward of that child) is bias-corrected with the group's count, not with its own.
"""
import ctypes

import torch
from torch.optim.optimizer import Optimizer

from ...hip import convnet, lib


class _Owner(object):
    """identity of one parameter group towards hip/convnet.params_updated (caches which banks hold its tensors)"""


class HipAdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = list(params)
        dev = next((p.device for p in params if torch.is_tensor(p)), torch.device('cpu'))
        lr_t = lr if torch.is_tensor(lr) else torch.tensor(float(lr), dtype=torch.float32, device=dev)
        super().__init__(params, dict(lr=lr_t, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._built = None          # (signature of gradient pointers, tables) per group
        self._owners = {}
        self.grad_norm = None

    # -- flat state ---------------------------------------------------------------------------
    def _ensure_state(self, group):
        ps = [p for p in group['params'] if p.requires_grad]
        if not ps:                  # a child frozen entirely (config.freeze / the optimizer's parameter regex): no-op
            return ps
        if 'flat' in group and group['flat']['n'] == sum(p.numel() for p in ps):
            return ps
        dev = ps[0].device
        n = sum(p.numel() for p in ps)
        flat = dict(n=n, m=torch.zeros(n, dtype=torch.float32, device=dev), v=torch.zeros(n, dtype=torch.float32, device=dev),
                    step=torch.zeros(1, dtype=torch.float32, device=dev),
                    norm_coef=torch.zeros(2, dtype=torch.float32, device=dev))
        off = 0
        for p in ps:
            assert p.dtype == torch.float32 and p.is_contiguous(), 'HipAdamW updates contiguous fp32 parameters'
            k = p.numel()
            old = self.state.get(p)
            st = {'step': flat['step'].view(()), 'exp_avg': flat['m'][off:off + k].view_as(p),
                  'exp_avg_sq': flat['v'][off:off + k].view_as(p)}
            if old:                                   # state loaded from a checkpoint: adopt its values
                st['exp_avg'].copy_(old['exp_avg'])
                st['exp_avg_sq'].copy_(old['exp_avg_sq'])
                flat['step'].fill_(float(old['step']))
            self.state[p] = st
            off += k
        group['flat'] = flat
        return ps

    def _table(self, group, ps, pin=None):
        """device table (+ partial-sum buffer) over the parameters that have a gradient now.  Tables are keyed by the
        gradient-storage signature.  ``pin`` (a tag, from ``prepare``): the table belongs to a captured step -- graphs of
        different phases (warm-up / GAN) each replay with the table of THEIR static gradient tensors, and a table a graph
        points at must outlive it -- and stays until ``release(tag)``.  Eager steps (``Optimizer.zero_grad`` sets gradients
        to None, so autograd allocates new ones every step and the signature keeps changing) share ONE replaceable slot:
        a 200k-step eager run holds one table, not one per step."""
        if not ps:
            return None, None, 0, 0
        live = [p for p in ps if p.grad is not None]
        sig = tuple((p.data_ptr(), p.grad.data_ptr()) for p in live)
        tables = group.setdefault('tables', {})              # signature -> [entry, tags]
        hit = tables.get(sig)
        if hit is not None:
            if pin is not None:
                hit[1].add(pin)
            return hit[0]
        slot = group.get('eager_table')
        if slot is not None and slot[0] == sig:
            if pin is None:
                return slot[1]
            tables[sig] = [slot[1], {pin}]                   # (an eager step's table a capture is about to point at)
            group['eager_table'] = None
            return slot[1]
        chunk = lib.get().msmc_opt_chunk()
        items = (lib.OptTensor * max(1, len(live)))()
        blocks = 0
        for it, p in zip(items, live):
            g = p.grad
            assert g.dtype == torch.float32 and g.is_contiguous(), 'HipAdamW takes contiguous fp32 gradients'
            st = self.state[p]
            it.p, it.g, it.m, it.v = p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
            it.n, it.first_chunk = p.numel(), blocks
            blocks += (p.numel() + chunk - 1) // chunk
        dev = ps[0].device
        if dev.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('HipAdamW: the tensor table must exist before hipGraph capture (gradient storage changed): '
                               'call optimizer.prepare() between the backward pass and the captured step')
        table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(dev)
        partial = torch.empty(max(1, blocks), dtype=torch.float32, device=dev)
        entry = (table, partial, len(live), blocks)
        if pin is not None:
            tables[sig] = [entry, {pin}]
        else:
            group['eager_table'] = (sig, entry)              # replaces the previous eager table (stream-ordered release)
        return entry

    def prepare(self, tag='graph'):
        """build the flat state and the tensor table for the gradients that exist now (before capturing a step); the
        table is pinned under ``tag`` until ``release(tag)``"""
        for group in self.param_groups:
            self._table(group, self._ensure_state(group), pin=tag)

    def release(self, tag='graph'):
        """the captured steps that pinned their tables under ``tag`` are gone: drop the tables nothing else pins"""
        for group in self.param_groups:
            tables = group.get('tables', {})
            for sig in [s for s, (_, tags) in tables.items() if tag in tags]:
                tables[sig][1].discard(tag)
                if not tables[sig][1]:
                    del tables[sig]

    @torch.no_grad()
    def step(self, closure=None, max_norm=0.0):
        """one AdamW step of every group; ``max_norm > 0`` clips the global gradient norm of the group first (in place,
        like clip_grad_norm_) -- ``self.grad_norm`` then holds the pre-clip norm (device tensor)"""
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            ps = self._ensure_state(group)
            table, partial, nt, blocks = self._table(group, ps)
            if nt == 0:
                continue
            flat = group['flat']
            lr = group['lr']
            if not torch.is_tensor(lr):
                lr = group['lr'] = torch.tensor(float(lr), dtype=torch.float32, device=ps[0].device)
            b1, b2 = group['betas']
            lib.check(lib.get().msmc_opt_clip_adamw(lib.ptr(table), nt, blocks, float(max_norm or 0.0), lib.ptr(partial),
                                                    lib.ptr(flat['norm_coef']), lib.ptr(lr.reshape(1)), lib.ptr(flat['step']),
                                                    float(b1), float(b2), float(group['eps']), float(group['weight_decay']), 1,
                                                    lib.stream(partial)), 'msmc_opt_clip_adamw')
            self.grad_norm = flat['norm_coef'][0]
            convnet.params_updated(self._owners.setdefault(id(group), _Owner()), ps)    # (raw-pointer update: no version counter moves)
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            group.pop('flat', None)
            group.pop('tables', None)
            group.pop('eager_table', None)
            if not torch.is_tensor(group['lr']):
                dev = group['params'][0].device
                group['lr'] = torch.tensor(float(group['lr']), dtype=torch.float32, device=dev)
            # re-home the loaded moments in the flat buffers NOW: a later re-homing (first step) would replace the state
            # tensors, and anything that recorded them in between -- the trainer's capture snapshot / roll-back -- would
            # restore into orphans and zero the live moments (resume + --graphs lost step / exp_avg that way)
            self._ensure_state(group)

    def state_dict(self):
        sd = super().state_dict()
        for g in sd['param_groups']:
            g.pop('flat', None)
            g.pop('tables', None)
            g.pop('eager_table', None)
            if torch.is_tensor(g['lr']):
                g['lr'] = float(g['lr'])
        for st in sd['state'].values():           # torch.optim.AdamW layout: a 0-dim fp32 ``step`` per parameter
            st['step'] = st['step'].detach().clone().reshape(())
            st['exp_avg'] = st['exp_avg'].detach().clone()
            st['exp_avg_sq'] = st['exp_avg_sq'].detach().clone()
        return sd
