"""Mirror of the reference `dust3r/optim_factory.py:9-14`."""


def adjust_learning_rate_by_lr(optimizer, lr):
    for group in optimizer.param_groups:
        group['lr'] = lr * group['lr_scale'] if 'lr_scale' in group else lr
