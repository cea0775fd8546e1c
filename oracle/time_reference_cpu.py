"""Time the REAL reference Trainer (imported unmodified from /root/reference through oracle/refshim, as
oracle/make_golden.py does) beside the oracle's restatement on THIS container's CPU cores: same model
(ResNet-50, fp32), same batch, same thread count, same step (forward + backward + SGD through Trainer.train).

Test infrastructure - not importable from the product.  The GPU box has no /root/reference, so bench.py's
`cpu_baseline` can only time the oracle there (kind "port"); this script is the evidence that the port runs at the
reference's own CPU speed, i.e. that the port is a fair stand-in for SURVEY.md section 8(d)'s reference CPU path.
Writes tests/golden/reference_cpu_timing.json.

    python oracle/time_reference_cpu.py [batch] [steps]
"""
import json
import os
import sys
import time

sys.dont_write_bytecode = True      # /root/reference is read-only: importing it must not leave __pycache__ behind

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get('CONVNET_REFERENCE', '/root/reference')

_T_START = __import__('time').time()


def assert_no_new_bytecode():
    """/root/reference is read-only: nothing this process imports from it may leave bytecode behind.  (Round 3's
    oracle/time_reference_cpu.py ran without sys.dont_write_bytecode and left *.pyc there, stamped 2026-09-26 18:25;
    this tree does not own the reference and cannot delete them, so only files written since this process started count.)"""
    for d, _, files in os.walk(REF):
        if os.path.basename(d) == '__pycache__':
            new = [f for f in files if os.path.getmtime(os.path.join(d, f)) >= _T_START - 1.0]
            assert not new, 'bytecode leaked into the reference tree: %s/%s' % (d, new[:3])

sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, 'refshim'))

import torch  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    import models as ref_models                      # the reference's registry (trainer.py:179-263 drives it)
    from trainer import Trainer as RefTrainer
    from utils.optim import OptimRegime
    from utils.cross_entropy import CrossEntropyLoss
    sys.path.insert(0, ROOT)
    from oracle import convnet_oracle as O
    threads = O.usable_cpus()
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(7)
    data = [(torch.randn(B, 3, 224, 224, generator=g), torch.randint(0, 1000, (B,), generator=g))
            for _ in range(steps + 1)]

    torch.manual_seed(123)
    model = ref_models.resnet(dataset='imagenet', depth=50)
    tr = RefTrainer(model, CrossEntropyLoss(), OptimRegime(model, model.regime), device_ids=None, device='cpu',
                    dtype=torch.float, distributed=False, print_freq=10 ** 9)
    tr.train([data[0]])                              # warm-up step
    t0 = time.perf_counter()
    for x, t in data[1:]:
        tr.train([(x, t)])
    ref_s = (time.perf_counter() - t0) / steps

    r = O.time_cpu_baseline(depth=50, batch=B, steps=steps, warmup=1, size=224)
    out = {'model': 'ResNet-50 fp32 3x224x224', 'batch': B, 'steps': steps, 'threads': threads,
           'reference_trainer_s_per_step': round(ref_s, 3), 'reference_trainer_img_s': round(B / ref_s, 2),
           'oracle_img_s': round(float(r['img_per_s']), 2), 'oracle_over_reference': round(float(r['img_per_s']) / (B / ref_s), 3),
           'torch': torch.__version__, 'note': 'build container CPU; the GPU box times the oracle only (bench.py cpu_baseline)'}
    with open(os.path.join(ROOT, 'tests', 'golden', 'reference_cpu_timing.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out))
    assert_no_new_bytecode()


if __name__ == '__main__':
    main()
