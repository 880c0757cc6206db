"""Constant-then-exponential learning-rate decay (reference lr_schedulers/exponential_lr.py:4-30):
lr = max(final, base * decay ** ((step - warmup) / scale)) once step >= warmup."""


class ExponentialDecayLRScheduler(object):
    def __init__(self, warmup_steps=50000, decay_scale=50000, decay_learning_rate=0.5, final_learning_rate=1e-5):
        self.warmup_steps, self.decay_scale = warmup_steps, decay_scale
        self.decay_learning_rate, self.final_learning_rate = decay_learning_rate, final_learning_rate

    def get_scale(self, steps):
        if steps < self.warmup_steps:
            return 1.0
        return float(self.decay_learning_rate) ** ((steps - self.warmup_steps) / self.decay_scale)

    def step(self, optimizer, steps):
        scale = self.get_scale(steps)
        for key, opt in optimizer.optimizers.items():
            lr = max(self.final_learning_rate, scale * optimizer.config[key].learning_rate)
            for group in opt.param_groups:
                if hasattr(group['lr'], 'fill_'):       # capturable optimizers keep lr in a device tensor
                    group['lr'].fill_(lr)
                else:
                    group['lr'] = lr
