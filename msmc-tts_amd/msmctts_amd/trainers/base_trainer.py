"""Training loop shell (re-expression of reference msmctts/trainers/base_trainer.py:16-142).

Keeps the reference's observable contract -- per-iteration order (LR schedule, batch to device,
``model.zero_grad``, ``train_step``, checkpoint cadence), checkpoint dictionary
``{'model', 'optimizer', 'iteration', 'config'}`` with the reference's key names, resume rules -- while
``train()`` builds its loader from the configuration's ``dataset:`` section (``msmctts_amd.datasets``: the
reference's id-list / feature-path conventions and collation, behind a ``DeviceLoader`` that keeps one batch ahead on
the GPU) or takes any iterable of collated batches (the metric uses the synthetic generator in
``msmctts_amd.synthetic``).
"""
import glob
import os
import re

import torch

from ..distributed.distributed import apply_gradient_allreduce
from ..utils.utils import load_checkpoint, to_model
from .lr_schedulers import build_lr_scheduler
from .optimizers import build_optimizer


class BaseTrainer(object):
    def __init__(self, config, model, num_gpus=1, rank=0):
        self.config = config
        self.distributed = num_gpus > 1
        self.rank = rank
        if hasattr(config, 'freeze') and config.freeze != '':
            for name, p in model.named_parameters():
                if re.match(config.freeze, name):
                    p.requires_grad = False
        if num_gpus > 0 and torch.cuda.is_available():      # without a GPU the first HIP op raises (no CPU path)
            model = model.cuda()
        if self.distributed:
            model = apply_gradient_allreduce(model)
        self.model = model
        self.optimizer = None

    # -- hooks -----------------------------------------------------------------------------
    def train_step(self, batch, iteration):
        raise NotImplementedError

    def replays(self, iteration):
        """True when ``train_step(iteration)`` replays captured hipGraphs (trainers with a graph mode override)."""
        return False

    def _sync_grads(self):
        reducer = getattr(self.model, 'grad_reducer', None)
        if reducer is not None:
            reducer.finish()

    # -- loop ------------------------------------------------------------------------------
    def train(self, data_loader=None, logger=None):
        """Run until ``config.training_steps``; ``data_loader`` is re-iterated as epochs.  Without one the loader is built
        from ``config.dataset`` / ``config.dataloader`` like the reference's loop does (base_trainer.py:40-41), sharded per
        rank under data parallelism, and wrapped in a ``DeviceLoader`` (one batch ahead on the GPU; with ``use_graphs``
        and a fixed ``segment_length`` every batch is padded to that many frames, i.e. static shapes)."""
        graphs = getattr(self, 'use_graphs', False)
        sampler = None
        if data_loader is None:
            from ..datasets import DeviceLoader, build_dataloader
            dataset, sampler, host_loader = build_dataloader(self.config.dataset, self.config.dataloader,
                                                             bool(getattr(self, 'distributed', False)))
            device = next(self.model.parameters()).device
            hop = getattr(dataset, 'frameshift', {}).get('mel')
            seg = getattr(dataset, 'segment_length', -1)
            pad = seg // hop if (graphs and hop and seg and seg > 0) else None
            data_loader = DeviceLoader(host_loader, device, pad_frames=pad, hop=hop,
                                       mel_pad=getattr(dataset, 'padding_value', {}).get('mel', 0.0))
        if self.optimizer is None:
            self.optimizer = build_optimizer(self.model, self.config.optimizer, capturable=graphs)
        scheduler = build_lr_scheduler(self.config.lr_scheduler) if 'lr_scheduler' in self.config else None
        iteration = self.attempt_load_checkpoint()
        self.model.train()
        epoch = 0
        while True:
            if sampler is not None:
                sampler.set_epoch(epoch)
            epoch += 1
            for batch in data_loader:
                if scheduler is not None:
                    scheduler.step(self.optimizer, iteration)
                batch = to_model(batch)
                if not (graphs and self.replays(iteration)):    # replayed steps own static gradient buffers
                    self.model.zero_grad()
                    self.optimizer.zero_grad()
                log = self.train_step(batch, iteration)
                if logger is not None:
                    logger(iteration, log)
                if self.rank == 0 and iteration > 0 and iteration % self.config.iters_per_checkpoint == 0:
                    self.save_checkpoint('{}/model_{}'.format(self.config.save_checkpoint_dir, iteration), iteration)
                if iteration >= self.config.training_steps:
                    return iteration
                iteration += 1

    # -- checkpoints -----------------------------------------------------------------------
    def attempt_load_checkpoint(self):
        restore = self.config.restore_checkpoint_path
        latest = self.find_latest_checkpoint()
        if self.config.resume_training and latest != '':
            restore = latest
        if restore != '':
            return load_checkpoint(restore, self.model, self.optimizer) + 1
        if self.config.pretrain_checkpoint_path != '':
            load_checkpoint(self.config.pretrain_checkpoint_path, self.model)
        return 0

    def find_latest_checkpoint(self):
        directory = self.config.save_checkpoint_dir
        if not directory or not os.path.exists(directory):
            return ''
        steps = [int(p.split('_')[-1]) for p in glob.glob(os.path.join(directory, 'model_*'))]
        if not steps or max(steps) == 0:
            return ''
        return os.path.join(directory, 'model_' + str(max(steps)))

    def save_checkpoint(self, filepath, iteration):
        os.makedirs(os.path.dirname(filepath) or '.', exist_ok=True)
        torch.save({'model': self.model.state_dict(), 'optimizer': self.optimizer.state_dict(),
                    'iteration': iteration, 'config': self.config.to_dict()}, filepath)
