#!/usr/bin/env python
"""Soak of the launch plan at the headline size: ResNet-50 bf16 b=256, N steps with eager launches and N steps under the plan
from the same seed and the same batches (an lr change in the middle): every reported loss, the final master weights, the
momentum buffers and the BatchNorm statistics must be bit-identical.
    python tools/soak_plan.py [steps=200]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch          # noqa: E402
import convnet_amd as ca   # noqa: E402


def run(plan, steps, pool):
    torch.manual_seed(123)
    model = ca.models.resnet(dataset='imagenet', depth=50)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(smooth_eps=0.1), ca.OptimRegime(model, model.regime), device='cuda:0',
                    dtype=torch.bfloat16, print_freq=10 ** 9)
    tr._graph_mode, tr._use_graph, tr._plan = ('auto', True, True) if plan else ('0', False, False)
    losses = []
    t0 = time.time()
    for i in range(0, steps, 10):
        if i == steps // 2:
            tr.epoch = 30          # models/resnet.py:253: lr 0.1 -> 0.01
        r = tr.train([pool[(i + j) % len(pool)] for j in range(10)])
        losses.append((r['loss'], r['prec1'], r['prec5']))
    torch.cuda.synchronize()
    dt = time.time() - t0
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    state['__momentum'] = tr.optimizer.momentum_buf.detach().clone()
    replays = sum(g['graph']['plan'].info()[7] for g in tr._gstates.values() if g.get('graph') and g['graph'].get('plan'))
    return losses, state, dt, replays


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    g = torch.Generator().manual_seed(7)
    pool = [(torch.randn(256, 3, 224, 224, generator=g).cuda(), torch.randint(0, 1000, (256,), generator=g).cuda()) for _ in range(4)]
    le, se, te, _ = run(False, steps, pool)
    lp, sp, tp, replays = run(True, steps, pool)
    same_loss = le == lp
    diff = [k for k in se if not torch.equal(se[k], sp[k])]
    print('%d steps, ResNet-50 bf16 b=256, label smoothing 0.1, lr change at step %d' % (steps, steps // 2))
    print('eager: %.2f s (%.0f img/s incl. a report sync every 10 steps)   plan: %.2f s (%.0f img/s), %d replays' %
          (te, steps * 256 / te, tp, steps * 256 / tp, replays))
    print('losses / prec@1 / prec@5 of the %d report points identical: %s   (first %s, last %s)' % (len(le), same_loss, le[0], le[-1]))
    print('state tensors compared: %d (weights, BatchNorm statistics, momentum); differing: %d %s' % (len(se), len(diff), diff[:5]))
    if not same_loss or diff:
        raise SystemExit(1)
    print('SOAK_OK')


if __name__ == '__main__':
    main()
