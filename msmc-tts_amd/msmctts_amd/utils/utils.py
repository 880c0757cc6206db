"""Small host helpers shared by the hot path (masks, plugin resolution, checkpoint loading).

Mirrors the parts of reference msmctts/utils/utils.py that the train step touches:
``get_mask_from_lengths`` (:154-158), ``module_search`` (:276-316), ``load_checkpoint`` (:207-250),
``to_model`` (:137-151).  Feature IO (npy / wav readers) lives in msmctts_amd/datasets/readers.py.
"""
import glob
import importlib
import inspect
import os
import re

import torch


def get_mask_from_lengths(lengths, max_len=None):
    """Bool mask, True on padding (t >= length)."""
    max_len = int(torch.max(lengths).item()) if max_len is None else max_len
    steps = torch.arange(0, max_len, device=lengths.device)
    return steps.unsqueeze(0) >= lengths.unsqueeze(1)


def to_model(batch, device=None):
    """Recursively move a collated batch to the training device (pinned + non_blocking when possible)."""
    if isinstance(batch, (list, tuple)):
        return [to_model(b, device) for b in batch]
    if isinstance(batch, dict):
        # ('*_host' entries are host-side copies kept on purpose, e.g. DeviceLoader's mel_length_host)
        return {k: (v if k.endswith('_host') else to_model(v, device)) for k, v in batch.items()}
    t = torch.as_tensor(batch).contiguous()
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else t.device
    return t.to(device, non_blocking=True)


def module_search(names, directory, package=None):
    """Resolve class name(s) by scanning ``directory/*.py`` and ``directory/*/__init__.py``.

    A dotted name ``pkg.Cls`` restricts the search to ``package.pkg``.  The same class found in two
    different files raises; a name found nowhere raises ``RuntimeError`` (reference behaviour).
    """
    wanted = [names] if isinstance(names, str) else list(names)
    files = glob.glob(os.path.join(directory, '*.py')) + glob.glob(os.path.join(directory, '*', '__init__.py'))
    mods = []
    for path in sorted(files):
        rel = os.path.relpath(path, directory)[:-3].replace(os.path.sep, '.')
        rel = rel[:-len('.__init__')] if rel.endswith('.__init__') else rel
        if rel and rel != '__init__':
            mods.append(rel)
    found = []
    for name in wanted:
        cls_name = name.split('.')[-1]
        prefix = name[:-len(cls_name) - 1]
        space = [prefix] if prefix else mods
        hit = None
        for rel in space:
            module = importlib.import_module('%s.%s' % (package, rel) if package else rel)
            obj = getattr(module, cls_name, None)
            if obj is None:
                continue
            if hit is not None and inspect.getfile(hit) != inspect.getfile(obj):
                raise RuntimeError('Repeated Module for %s: %s, %s' % (cls_name, inspect.getfile(hit),
                                                                       inspect.getfile(obj)))
            hit = obj if hit is None else hit
        if hit is None:
            raise RuntimeError('Found dismatched modules for %s' % (names,))
        found.append(hit)
    return found[0] if isinstance(names, str) else found


def load_checkpoint(source, model, optimizer=None, module=None):
    """Checkpoint dict/path/list-of-(regex, path) loader; returns the stored iteration."""
    if isinstance(source, (list, tuple)):
        return max([0] + [load_checkpoint(obj, model, optimizer, pattern) for pattern, obj in source])
    if isinstance(source, str):
        if not os.path.isfile(source):
            raise AssertionError('checkpoint not found: %s' % source)
        ckpt = torch.load(source, map_location='cpu', weights_only=False)
    elif isinstance(source, dict):
        ckpt = source
    else:
        raise TypeError('Unacceptable type: %s' % type(source))
    weights = ckpt['model']
    if module is not None:
        keep = {k: weights[k] for k in model.state_dict() if re.match(module, k)}
        model.load_state_dict(keep, strict=False)
    else:
        try:
            model.load_state_dict(weights)
            if optimizer is not None:
                optimizer.load_state_dict(ckpt['optimizer'])
        except Exception:
            print('Loaded model is not the same as the current one')
            model.load_state_dict(weights, strict=False)
    return ckpt.get('iteration', 0)
