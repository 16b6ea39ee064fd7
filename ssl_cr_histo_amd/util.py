"""util.py:26-46 of the reference -- the only piece of util.py on the hot path."""


class AverageMeter(object):
    """Computes and stores the average and current value (weighted running mean)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
