#!/usr/bin/env python
"""Per-tensor step-0 gradient errors of the engine against the warm-start goldens (GPU)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import load_warm, rel_l2, run_engine_trajectory   # noqa: E402

tags = sys.argv[1].split(',') if len(sys.argv) > 1 else ['r18_b256_warm', 'r50_b256_warm']
for tag in tags:
    meta, tens = load_warm(tag)
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        grads = {k: None for k in tens['grad0']}
        recs, tr, model, data = run_engine_trajectory(meta, dtype, torch.device('cuda', 0), steps=2, grads_after_step0=grads)
        res[dtype] = (recs, grads)
        print(tag, dtype, 'records', [(round(r['loss'], 5), round(r['grad'], 5)) for r in recs], 'golden',
              [(round(r['loss'], 5), round(r['grad'], 5)) for r in meta['records'][:2]])
    for k, g in tens['grad0'].items():
        n32, v32 = res[torch.float32][1][k]
        n16, v16 = res[torch.bfloat16][1][k]
        print('%-34s gold norm %.4e | fp32 norm err %.1e l2 %.1e | bf16 norm err %.1e l2 %.1e | bf16 vs fp32 engine l2 %.1e' % (
            k, g['norm'], abs(n32 - g['norm']) / g['norm'], rel_l2(v32, g['val']), abs(n16 - g['norm']) / g['norm'],
            rel_l2(v16, g['val']), rel_l2(v16, v32)))
