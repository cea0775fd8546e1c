"""AverageMeter / accuracy with the semantics Trainer relies on (/root/reference trainer.py:7,
181-184,224-229; the classes live in the un-vendored utils.meters and are re-stated from their
call sites): ``update(val, n)`` keeps val, sum += val*n, count += n, avg = sum/count; accuracy
returns prec@k in percent of the batch."""
from . import ops


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0.
        self.avg = 0.
        self.sum = 0.
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count if self.count else 0.

    def set(self, val, total, count):
        """Install values accumulated on the device (sum and count) plus the latest value."""
        self.val = val
        self.sum = total
        self.count = count
        self.avg = total / count if count else 0.


def accuracy(output, target, topk=(1,)):
    """prec@k (%) of `output` logits vs integer `target`, computed by the HIP kernel; returns a list
    of 0-dim device tensors (no host sync), only k in {1, 5} like the reference's call."""
    stats = ops.accuracy_counts(output, target)
    res = []
    for k in topk:
        if k == 1:
            res.append(stats[1])
        elif k == 5:
            res.append(stats[2])
        else:
            raise NotImplementedError('accuracy(): only top-1 / top-5 are computed on the device')
    return res
