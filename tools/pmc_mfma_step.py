#!/usr/bin/env python
"""MFMA utilisation per kernel of the ResNet-50 step from hardware counters: one rocprofv3 --pmc pass (SQ_INSTS_MFMA,
SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES; kernel trace only) over `bench.py --steps 2 --warmup 1`, joined
with the launch durations of the same pass.  Per kernel and for the whole step:
  mfma / peak = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)   (the counter counts SIMD-cycles: 32 per
                v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md) - the fraction of the dense peak the matrix pipes were busy,
                independent of bench.py's algorithmic FLOP accounting.  Counter collection serialises the launches: the
                durations are those of each kernel ALONE (their sum exceeds the two-stream step).

    python tools/pmc_mfma_step.py OUTDIR [STEP_MS]   (GPU box; writes OUTDIR/mfma_step.txt; STEP_MS = the untraced step time)"""
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CTRS = 'SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES'


def main():
    out = os.path.abspath(sys.argv[1])
    os.makedirs(out, exist_ok=True)
    d = os.path.join(out, 'sqstep')
    env = dict(os.environ, TMPDIR='/tmp', CONVNET_AMD_FLAGS='graph=0')
    cmd = ['rocprofv3', '--pmc'] + CTRS.split() + ['--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'sq', '--',
                                                   sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1',
                                                   '--no-cpu-baseline', '--no-kernel-profile']
    r = subprocess.run(cmd, env=env, cwd='/tmp', stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    if r.returncode != 0:
        print(r.stdout[-2000:])
        raise SystemExit('rocprofv3 pass failed')
    ctr = defaultdict(lambda: defaultdict(float))      # dispatch id -> counter -> value
    name = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            ctr[row['Dispatch_Id']][row['Counter_Name']] += float(row['Counter_Value'])
            name[row['Dispatch_Id']] = row['Kernel_Name']
    dur = {}
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row['Dispatch_Id']] = int(row['End_Timestamp']) - int(row['Start_Timestamp'])
    agg = defaultdict(lambda: defaultdict(float))
    for did, c in ctr.items():
        k = name[did].split('(')[0].replace('void ', '')
        a = agg[k]
        a['n'] += 1
        a['ns'] += dur.get(did, 0)
        for cn, v in c.items():
            a[cn] += v
    # the number of profiled steps is read off the trace (a kernel that runs exactly once per training step), the step time
    # the whole-step figure is priced on comes from the caller (argv[2], ms: the untraced bench line of the same tree)
    once = [a['n'] for k, a in agg.items() if k.startswith('nchw_to_pairs_kernel') or k.startswith('nchw_to_nhwc_kernel')]
    steps = float(once[0]) if once and once[0] > 0 else 3.0
    step_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 17.25
    CLOCK_GHZ = 2.4       # the guide's peak clock; under dense MFMA the part sustains ~1.9 GHz (NOTES.md), so 'of the dense peak'
    #                       below is against the DATASHEET peak, as everywhere in bench.py
    lines = ['MFMA utilisation per kernel from SQ counters (rocprofv3 --pmc %s; %d profiled steps counted in the trace, eager, two streams; '
             'peak priced at %.1f GHz x 1024 SIMDs):' % (CTRS, int(steps), CLOCK_GHZ),
             '%-78s %7s %9s %11s %14s' % ('kernel', 'n/step', 'ms/step', 'mfma/peak', 'MFMA insts/step')]
    tot_busy = tot_ns = 0.0
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['ns']):
        if a['ns'] <= 0:
            continue
        peak_cycles = a['ns'] * CLOCK_GHZ * 1024.0
        lines.append('%-78s %7.1f %9.3f %11.3f %14.3e' % (k[:78], a['n'] / steps, a['ns'] / steps / 1e6,
                                                          a['SQ_VALU_MFMA_BUSY_CYCLES'] / peak_cycles,
                                                          a['SQ_INSTS_MFMA'] / steps))
        tot_busy += a['SQ_VALU_MFMA_BUSY_CYCLES']
        tot_ns += a['ns']
    lines.append('')
    lines.append('sum of kernel durations %.2f ms/step (serialised by the counter pass); MFMA-busy SIMD-cycles per step %.3e '
                 '= %.3f of the dense peak over a %.2f ms step' % (tot_ns / steps / 1e6, tot_busy / steps,
                                                                 tot_busy / steps / (step_ms * 1e6 * CLOCK_GHZ * 1024.0), step_ms))
    txt = '\n'.join(lines)
    open(os.path.join(out, 'mfma_step.txt'), 'w').write(txt + '\n')
    print('\n'.join(lines[:28]))
    print(lines[-1])


if __name__ == '__main__':
    main()
