#!/usr/bin/env python
"""bench.py -- ResNet-50 bf16 training throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (Trainer._step: NCHW fp32 -> NHWC bf16 conversion, forward,
softmax-CE + accuracy meters, backward, [RCCL all-reduce], fused SGD+momentum) over one synthetic
batch of 256 images 3x224x224 per GPU.  Inputs are resident in HBM when the timed region starts
(a pool of pre-staged device batches is cycled; the reference's per-step H2D copy is excluded).
Weak scaling: per-GPU batch fixed, value = images of all ranks / max-over-ranks time.

Rank 0 prints ONE compact JSON line (the contract object, < 4 KB, the LAST line of stdout).  The per-kernel and
per-layer tables (`kernels`, `kernels_overlapped`, `conv_layers`) go to a detail file (`--detail-out`, default
gpurun_out/bench_detail.json; `--detail-stdout` prints them as an EARLIER {"detail": ...} line).  Objects:
  roofline     : live HIP-event timing (a separate profiled pass after the timed region) of the
                 dominant kernel family: algorithmic FLOPs (2*MACs of
                 conv/fc fwd+dgrad+wgrad) or bytes per launch / measured launch time vs the
                 gfx950 peak (MI355X_MICROARCH.md: 2.5 PFLOP/s dense bf16 MFMA, 8 TB/s HBM3E).
  kernels      : the same figures for every kernel family of the step (time share per step).
  conv_layers  : per conv layer shape and direction: us, TFLOP/s, GB/s, which roof binds, fraction of it.
  kernels_overlapped : the same launches timed in a pass that keeps the step's two-stream schedule (events on the
                 launching stream): the durations rocprofv3 reports for the timed region.  `roofline` names the kernel
                 with the largest total there and prices it there (`roofline.alone`: the same launches by themselves).
  roofline.traffic : HBM bytes per launch of that kernel from rocprofv3 PMC passes - measured by this command with
                 `--pmc` (two extra passes of a short run, tagged live), else quoted from the latest committed passes
                 (tagged static).
  hbm_measured_whole_step : HBM traffic of ALL kernels of a step (the same PMC source) over this run's step time: the
                 step as a whole against the memory system.
  cpu_baseline : the CPU oracle (oracle/convnet_oracle.py, kind "port": the reference tree is not on the GPU box)
                 timed on this host's cores on a bounded sample (ResNet-50 fp32, batch 32, 1 warm-up + 5 steps).
"""
import argparse
import json
import os
import sys
import time

# the host driver supports dmabuf IPC only: without this RCCL / cross-process device buffers fail with
# "hipIpcGetMemHandle: invalid argument" (already exported on the GPU boxes; kept here for a bare launch)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {'bf16': 2500.0, 'f16': 2500.0, 'f32': 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
TRAIN_GFLOP_PER_IMG = {50: 24.2991, 18: 10.6484}
# compulsory HBM traffic per image of the SURVEY.md section 8(d) traffic model (MB): the whole-step HBM roofline
MODEL_MB_PER_IMG = {(50, 'bf16'): 347.7, (18, 'f32'): 151.4}


def _pmc_file_order(name):
    """profiles/rNN[x]_pmc_traffic.json: by round, the round's closing set (no letter) last."""
    import re
    m = re.match(r'r(\d+)([a-z]*)_', name)
    return (int(m.group(1)), m.group(2) == '', name) if m else (-1, False, name)


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


LIVE_PMC = {'pm': None}      # per-kernel traffic measured by this command (--pmc), shared by the two places that quote it


def roof_fraction(records, dtype):
    """sum_i max(bytes_i / HBM peak, flops_i / MFMA peak) / sum_i measured time_i over (ms, flops, bytes) records:
    how far a set of launches sits from whichever roof binds each of them."""
    ideal = sum(max(b / (PEAK_HBM_GBS * 1e9), f / (PEAK_TFLOPS[dtype] * 1e12)) for _, f, b in records)
    meas = sum(ms for ms, _, _ in records) * 1e-3
    return round(ideal / meas, 4) if meas > 0 else None


def run_pmc_passes(args):
    """roofline.traffic measured by THIS command: two rocprofv3 counter passes (FETCH_SIZE, then WRITE_SIZE - they do
    not fit one pass; kernel trace only, never combined with other trace domains) over a 1 warm-up + 2 step run of
    the same workload on eager launches, corrected as MI355X_MICROARCH.md prescribes (KiB units, FETCH_SIZE x2 on
    gfx950): tools/pmc_traffic.py.  Returns the per-kernel dict or None when rocprofv3 is not usable here."""
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import pmc_traffic
    if shutil.which('rocprofv3') is None:
        return None
    tmp = tempfile.mkdtemp(prefix='bench_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp', CONVNET_AMD_FLAGS=','.join(filter(None, [os.environ.get('CONVNET_AMD_FLAGS', ''), 'graph=0'])))
    cmd = [sys.executable, os.path.abspath(__file__), '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
           '--no-kernel-profile', '--no-issue-probe', '--batch', str(args.batch), '--depth', str(args.depth), '--dtype', args.dtype] + \
        (['--quantize'] if args.quantize else [])
    for name, ctr in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
        r = subprocess.run(['rocprofv3', '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d',
                            os.path.join(tmp, name), '-o', name, '--'] + cmd, env=env, cwd='/tmp',
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
        if r.returncode != 0:
            sys.stderr.write('bench.py --pmc: %s pass failed:\n%s\n' % (ctr, r.stdout[-1500:]))
            return None
    res = pmc_traffic.collect(tmp, steps=3)
    res['workload'] = 'bench.py --batch %d --depth %d --dtype %s%s' % (args.batch, args.depth, args.dtype,
                                                                      ' --quantize' if args.quantize else '')
    if args.pmc_out:
        with open(args.pmc_out, 'w') as f:
            json.dump(res, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch (BASELINE: 256)')
    ap.add_argument('--depth', type=int, default=50)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16', 'f32'])
    ap.add_argument('--pool', type=int, default=4, help='distinct pre-staged device batches')
    ap.add_argument('--host-inputs', action='store_true',
                    help='feed pinned HOST batches (PCIe-inclusive rate, for DESIGN.md; never the contract value)')
    ap.add_argument('--quantize', action='store_true',
                    help="BASELINE config 5: resnet(quantize=True), the simulated-8-bit operators (a parity-test "
                         "configuration, not the contract workload)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--no-issue-probe', action='store_true',
                    help='skip the three extra steps that time the host side of a step (the PMC sub-runs count steps)')
    ap.add_argument('--pmc', action='store_true',
                    help='also measure roofline.traffic LIVE: two extra rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; '
                         'kernel trace only) of a 3-step run of this same workload, ~2-4 min (N=1 only)')
    ap.add_argument('--pmc-out', default=None, help='where --pmc writes the per-kernel traffic JSON (for profiles/)')
    ap.add_argument('--detail-out', default=os.path.join(ROOT, 'gpurun_out', 'bench_detail.json'),
                    help='file for the per-kernel / per-layer tables (never part of the contract line)')
    ap.add_argument('--detail-stdout', action='store_true', help='also print the tables as an earlier {"detail": ...} line')
    ap.add_argument('--cpu-steps', type=int, default=15, help='timed steps of the CPU baseline (batch 32, 1 warm-up; ~1.7 s each)')
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU fallback for the measured path')
    share = os.environ.get('BENCH_SHARE_GPU') == '1'   # protocol test of the N > 1 flow on a 1-GPU box (gloo)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU,
        # the same command line the driver uses) and hand over to them
        if torch.cuda.device_count() < args.gpus and not share:
            raise SystemExit('bench.py: --gpus %d but only %d device(s) visible' % (args.gpus, torch.cuda.device_count()))
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    import convnet_amd as ca
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    distributed = world > 1 or os.environ.get('BENCH_FORCE_DIST') == '1'   # world-1 RCCL smoke test
    if share:
        local_rank = 0
        os.environ.setdefault('BENCH_DIST_BACKEND', 'gloo')   # RCCL refuses two ranks on one device
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit('bench.py: rank %d has no device (%d visible)' % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    rccl_log = None
    if distributed and 'NCCL_DEBUG' not in os.environ and os.environ.get('BENCH_DIST_BACKEND', 'nccl') == 'nccl':
        # what RCCL builds for this job (channels, rings / trees, transports) goes into the line: INFO-level INIT / GRAPH
        # messages into a per-process file (never stdout), read back after the communicators are up
        rccl_log = '/tmp/cn_bench_rccl_%d.log' % os.getpid()
        os.environ.update(NCCL_DEBUG='INFO', NCCL_DEBUG_SUBSYS='INIT,GRAPH', NCCL_DEBUG_FILE=rccl_log)
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        dist.init_process_group(backend=os.environ.get('BENCH_DIST_BACKEND', 'nccl'), init_method='env://',
                                world_size=world, rank=rank)
    assert not ca._lib.is_emulated()

    dtype = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[args.dtype]
    torch.manual_seed(123)
    model = ca.models.resnet(dataset='imagenet', depth=args.depth, quantize=args.quantize)
    crit = ca.CrossEntropyLoss()
    opt = ca.OptimRegime(model, model.regime)
    comm_note = None
    try:
        tr = ca.Trainer(model, crit, opt, device=str(device), dtype=dtype, distributed=distributed,
                        local_rank=local_rank, print_freq=10 ** 9)
    except ca._lib.ConvNetHipError as e:
        # The product has ONE transport and stops when it cannot be built (comm.py).  The bench alone recovers, loudly:
        # the set-up phases end in an agreement step, so every rank is here with the same verdict; the run continues
        # on torch.distributed's RCCL collectives and says so in the JSON line (`transport`) and on stderr.
        if not distributed or 'direct-RCCL set-up failed' not in str(e):
            raise
        comm_note = 'torch.distributed collectives; direct RCCL communicator failed: %s' % str(e)[:300]
        sys.stderr.write('bench.py[rank %d]: %s\n' % (rank, comm_note))
        os.environ['CONVNET_AMD_COMM'] = 'torch'
        torch.manual_seed(123)
        model = ca.models.resnet(dataset='imagenet', depth=args.depth, quantize=args.quantize)
        opt = ca.OptimRegime(model, model.regime)
        tr = ca.Trainer(model, crit, opt, device=str(device), dtype=dtype, distributed=distributed,
                        local_rank=local_rank, print_freq=10 ** 9)

    B = args.batch
    g = torch.Generator().manual_seed(123 + rank)
    pool = [(torch.randn(B, 3, 224, 224, generator=g), torch.randint(0, 1000, (B,), generator=g))
            for _ in range(args.pool)]
    if args.host_inputs:
        pool = [(x.pin_memory(), t.pin_memory()) for x, t in pool]
    else:
        pool = [(x.to(device), t.to(device)) for x, t in pool]

    def loader(n):
        return [pool[i % len(pool)] for i in range(n)]

    def fence():
        if distributed:
            # this rank's own bucket all-reduces (direct communicator, its own stream) are complete before the rank enters
            # the process group's barrier: two RCCL communicators never have kernels in flight on one device at once
            torch.cuda.synchronize()
            if dist.get_backend() == 'nccl':
                dist.barrier(device_ids=[local_rank])   # RCCL barrier on this rank's own device
            else:
                dist.barrier()
        torch.cuda.synchronize()

    t_gpu0 = time.perf_counter()
    tr.train(loader(args.warmup))          # W untimed warm-up steps
    fence()
    t0 = time.perf_counter()
    res = tr.train(loader(args.steps))     # EXACTLY K timed steps
    torch.cuda.synchronize()
    fence()
    elapsed = time.perf_counter() - t0
    # how the steps of the timed region were issued, and the host time one step costs (outside the timed region)
    modes = [('plan' if g['graph'].get('plan') is not None else 'hip-graph') for g in tr._gstates.values()
             if g.get('graph') is not None]
    with torch.cuda.stream(tr._main_stream) if tr._main_stream is not None else torch.cuda.stream(torch.cuda.current_stream(device)):
        tr.model.train()
        xb, tb = (t.to(device) for t in pool[0])
        host_ms = None
        for _ in range(0 if args.no_issue_probe else 3):          # ONE step issued into drained queues, three times (a deep backlog would make the host wait
            fence()                 # for queue space - that is device time, not issue time), the fastest counts
            th0 = time.perf_counter()
            tr._step(xb, tb, training=True)
            dt_ms = (time.perf_counter() - th0) * 1e3
            host_ms = dt_ms if host_ms is None else min(host_ms, dt_ms)
    fence()
    step_issue = {'mode': modes[0] if modes else 'eager', 'host_ms_per_step': round(host_ms, 2) if host_ms is not None else None}
    rank_devices, params_in_sync = [local_rank], None
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # which HIP device every rank ran on, by rank (one process per GPU: rank r -> device r on a full node)
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, torch.cuda.current_device())
        # data-parallel sanity of THIS run, outside the timed region: after W + K steps on different shards every rank must
        # hold the same master weights (same initial broadcast, same all-reduced gradients, same update) - a transport that
        # loses or reorders a bucket shows up here, not only in a loss curve
        p32 = tr.arena.params.double()
        sig = [None] * world
        dist.all_gather_object(sig, (float(p32.sum()), float(p32.abs().sum())))
        params_in_sync = all(abs(a - sig[0][0]) <= 1e-9 * abs(sig[0][0]) + 1e-12 and abs(b - sig[0][1]) <= 1e-9 * abs(sig[0][1])
                             for a, b in sig)

    # ---- live per-kernel timing: two profiled passes after the timed region, HIP events on the launch stream ----
    #  (1) overlapped: the step keeps its two-stream schedule; the events sit on the stream each call is launched on.
    #      These are the durations rocprofv3 reports for the timed region; the dominant kernel is chosen and priced here.
    #  (2) alone: the weight-gradient side stream folded into the main stream, every kernel runs by itself: the
    #      per-kernel / per-layer tables (how far each launch is from ITS roof without a neighbour sharing the chip).
    kernels, layers, roof, kernels_ovl = {}, {}, None, {}
    nprof = 2
    agg = agg_ovl = None
    if not args.no_kernel_profile:
        # every rank runs the profiled steps (they contain the gradient all-reduce: a rank-0-only pass would
        # wait for collectives the other ranks never enter); only rank 0 reports
        P = ca.ops.PROFILER
        P.records, P.enabled, P.fold_streams = [], True, False
        tr.train(loader(nprof))
        P.enabled = False
        fence()
        agg_ovl = P.summary() if rank == 0 else None
        P.records, P.enabled, P.fold_streams = [], True, True
        tr.train(loader(nprof))
        P.enabled = False
        fence()
    if rank == 0 and not args.no_kernel_profile:
        agg = ca.ops.PROFILER.summary()

        def table(ag):
            out, total_ms = {}, sum(a['ms'] for a in ag.values())
            for name, a in sorted(ag.items(), key=lambda kv: -kv[1]['ms']):
                sec = a['ms'] * 1e-3
                out[name] = {
                    'ms_per_step': round(a['ms'] / nprof, 3),
                    'share': round(a['ms'] / total_ms, 4),
                    'launches_per_step': a['launches'] // nprof,
                    'avg_us_per_launch': round(a['ms'] * 1e3 / max(a['launches'], 1), 2),
                    'tflops': round(a['flops'] / sec / 1e12, 2) if a['flops'] else None,
                    'gbs': round(a['bytes'] / sec / 1e9, 1),
                    # launches priced one by one against the roof that binds each (HBM-bound 1x1 and MFMA-bound 3x3
                    # layers share kernel names): sum of max(bytes/8 TB/s, flops/peak) over the measured time
                    'roof_frac': roof_fraction(a['records'], args.dtype),
                    # launched on the weight-gradient side stream (beside the backward chain, sharing the chip with it)
                    'side_stream': a.get('side_ms', 0.0) > 0.5 * a['ms'],
                }
            return out
        kernels = table(agg)
        kernels_ovl = {k: {f: v[f] for f in ('ms_per_step', 'launches_per_step', 'avg_us_per_launch', 'gbs', 'tflops',
                                             'roof_frac', 'side_stream')} for k, v in table(agg_ovl).items()}
        # the same per convolution layer shape (fwd / dgrad / wgrad): which roof binds it and how close it runs
        # (a weight gradient's fixed-order split reduction, timed separately, is folded back into its layer's row)
        det = ca.ops.PROFILER.summary(by_detail=True)
        for d in [d for d in det if d.endswith(' [reduce]')]:
            parent = det.get(d[:-len(' [reduce]')])
            if parent is not None:
                parent['ms'] += det[d]['ms']
                parent['bytes'] += det[d]['bytes']
            del det[d]
        for d, a in sorted(det.items(), key=lambda kv: -kv[1]['ms']):
            sec = a['ms'] * 1e-3
            t_hbm = a['bytes'] / (PEAK_HBM_GBS * 1e9)
            t_mfma = a['flops'] / (PEAK_TFLOPS[args.dtype] * 1e12)
            layers[d] = {'n': a['calls'] // nprof, 'us': round(a['ms'] * 1e3 / a['calls'], 1),
                         'tflops': round(a['flops'] / sec / 1e12, 1), 'gbs': round(a['bytes'] / sec / 1e9, 1),
                         'bound': 'mfma' if t_mfma >= t_hbm else 'hbm',
                         'roof_frac': round(max(t_hbm, t_mfma) / sec, 3)}
        # `roofline.kernel`: the SINGLE HIP kernel with the largest total time in the OVERLAPPED step, whichever stream it
        # is launched on (the weight-gradient side stream included), priced there; `alone` = the same launches with the
        # side stream folded in.  Call labels that name several kernels ('a+b+c': one BatchNorm-backward call = three
        # launches; 'x + y': a strided dgrad's parity classes) are never "the" kernel: the largest such call of the
        # main stream is listed beside it as `main_stream_call`.
        def single(label):
            base = label.split(' (')[0].split(' [')[0]
            depth, plus = 0, False
            for ch in base:
                depth += (ch == '<') - (ch == '>')
                plus = plus or (ch == '+' and depth == 0)
            return not plus

        def priced(lbl, tab, tab_alone):
            kk, ka = tab[lbl], tab_alone.get(lbl, {})
            if kk['tflops'] and kk['tflops'] / PEAK_TFLOPS[args.dtype] >= kk['gbs'] / PEAK_HBM_GBS:
                r = {'kernel': lbl, 'bound': 'mfma', 'achieved': kk['tflops'], 'peak': PEAK_TFLOPS[args.dtype],
                     'unit': 'TFLOP/s', 'frac': round(kk['tflops'] / PEAK_TFLOPS[args.dtype], 4)}
            else:
                r = {'kernel': lbl, 'bound': 'hbm', 'achieved': kk['gbs'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                     'frac': round(kk['gbs'] / PEAK_HBM_GBS, 4)}
            r.update({'avg_us_per_launch': kk['avg_us_per_launch'], 'launches_per_step': kk['launches_per_step'],
                      'ms_per_step': kk['ms_per_step'], 'side_stream': kk['side_stream'],
                      'frac_per_launch_roof': kk['roof_frac'],   # each launch against max(HBM, MFMA) time
                      'alone': {f: ka.get(f) for f in ('avg_us_per_launch', 'gbs', 'tflops', 'roof_frac')}})
            return r
        dom = next(k for k in kernels_ovl if single(k))
        roof = priced(dom, kernels_ovl, kernels)
        roof['traffic'] = None
        roof['algorithmic_bytes_per_launch'] = round(agg_ovl[dom]['bytes'] / max(agg_ovl[dom]['launches'], 1))
        roof['timing'] = ('HIP events on the launching stream around every launch, in a profiled pass that keeps the '
                          "timed region's two-stream schedule; alone = side stream folded in; rocprofv3: profiles/")
        main_call = next((k for k in kernels_ovl if not kernels_ovl[k]['side_stream']), None)
        if main_call is not None and main_call != dom:
            roof['main_stream_call'] = priced(main_call, kernels_ovl, kernels)
        pm, pm_file, live = None, None, False
        if args.pmc and world == 1:
            pm = run_pmc_passes(args)
            live = pm is not None
            LIVE_PMC['pm'] = pm
        if pm is None:
            # HBM traffic per launch NOT measured in this run (PMC counters need rocprofv3 passes of their own:
            # `--pmc`): the latest committed PMC result (profiles/*_pmc_traffic.json) is quoted and tagged static
            try:
                cands = sorted((f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('pmc_traffic.json')),
                               key=_pmc_file_order)
                pm_file = cands[-1]
                pm = json.load(open(os.path.join(ROOT, 'profiles', pm_file)))
            except Exception:
                pm = None
        if pm is not None:
            # a call label may name several kernels ('bn_bwd_reduce+bn_bwd_finalize+bn_bwd_apply': one BatchNorm-backward
            # call = three launches): the traffic per launch is the launch-weighted mean over the kernels it names
            base = dom.split(' (')[0].split(' [')[0]
            parts = [q.strip() for q in base.split('+')] if ('+' in base and '<' not in base) else [base]
            tot_b = tot_n = 0.0
            for part in parts:
                for kn, kv in pm['kernels'].items():
                    name = kn.replace('void ', '')
                    if name.startswith(part + '_kernel') or (len(parts) == 1 and name.startswith(part)):
                        tot_b += kv['hbm_bytes_per_launch'] * kv['launches']
                        tot_n += kv['launches']
            if tot_n > 0:
                roof['traffic'] = round(tot_b / tot_n)
                roof['traffic_unit'] = 'HBM bytes/launch, rocprofv3 --pmc FETCH_SIZE x2 (gfx950) + WRITE_SIZE, separate passes'
                roof['traffic_source'] = 'live: --pmc passes run by this command' if live else \
                    'static: profiles/%s' % pm_file
    gpu_active = time.perf_counter() - t_gpu0      # warm-up + timed region + issue probe + profiled passes (GPU busy throughout)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import convnet_oracle as O
        r = O.time_cpu_baseline(depth=args.depth, batch=32, steps=args.cpu_steps, warmup=1, size=224)
        cpu = {'value': round(r['img_per_s'], 2), 'unit': 'images/sec', 'cores': r['cores'], 'kind': 'port',
               'stat': 'median of %d steps' % args.cpu_steps,
               'spread': {'min': round(r['img_per_s_min'], 2), 'max': round(r['img_per_s_max'], 2),
                          'mean': round(r['img_per_s_mean'], 2)},
               'sample': 'oracle/convnet_oracle.py (CPU restatement of the reference Trainer step) ResNet-%d fp32 training, '
                         'batch 32, 1 warm-up + %d timed steps, each timed on its own (median %.2f s/step, %.0f s in all)'
                         % (args.depth, args.cpu_steps, r['s_per_step'], r['s_total'])}
        try:   # how the port compares with the REAL reference Trainer on the same cores (measured in the build container)
            with open(os.path.join(ROOT, 'tests', 'golden', 'reference_cpu_timing.json')) as f:
                rt = json.load(f)
            cpu['port_over_reference'] = rt['oracle_over_reference']
            cpu['port_over_reference_source'] = ('%d-core ratio, build container: reference Trainer %.2f vs port %.2f img/s '
                                                 '(oracle/time_reference_cpu.py, batch %d)'
                                                 % (rt['threads'], rt['reference_trainer_img_s'], rt['oracle_img_s'],
                                                    rt['batch']))
        except Exception:
            pass

    if rank == 0:
        img_s = B * world * args.steps / elapsed
        step_tflops = TRAIN_GFLOP_PER_IMG.get(args.depth, 0.0) * B * 1e-3
        out = {
            'metric': 'images/sec ResNet-%d %s%s 3x224x224 b=%d/GPU training (fwd+bwd+SGD)' % (
                args.depth, args.dtype, ' quantize=True (simulated 8-bit)' if args.quantize else '', B),
            'value': round(img_s, 1), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic' + (' (host batches, H2D inside the timed loop)' if args.host_inputs else ''),
            'config': {'workload': 'ResNet-%d (depth:%d) synthetic 3x224x224 b=%d/GPU %s, SGD+momentum, '
                                   '%s' % (args.depth, args.depth, B, args.dtype,
                                           'dp%d gradient all-reduce' % world if world > 1 else '1 MI355X'),
                       'global_batch': B * world, 'final_loss': round(float(res['loss']), 4),
                       'parallelism': 'dp%d' % world, 'rank_devices': rank_devices, 'params_in_sync_across_ranks': params_in_sync,
                       'transport': (tr.reducer.describe() if tr.reducer is not None else None) if comm_note is None
                       else '%s [%s]' % (tr.reducer.describe() if tr.reducer is not None else None, comm_note),
                       'rccl': ca.comm.topology_summary(rccl_log) if rccl_log else None},
            'mfma_frac_whole_step': round(step_tflops / (elapsed / args.steps) / PEAK_TFLOPS[args.dtype], 4)
            if step_tflops else None,
            'hbm_frac_whole_step': round(MODEL_MB_PER_IMG[(args.depth, args.dtype)] * 1e6 * B / (elapsed / args.steps)
                                         / (PEAK_HBM_GBS * 1e9), 4)
            if (args.depth, args.dtype) in MODEL_MB_PER_IMG and not args.quantize else None,
            # true when the run is NOT on the product's direct-RCCL transport (a DP number produced that way says
            # nothing about csrc/comm.hip)
            'transport_fallback': comm_note is not None,
            'roofline': roof, 'cpu_baseline': cpu,
            # which step the timed region ran (launch plan = csrc/plan.hip) and what it cost the host; which sources the
            # measured binary was built from (== the tree's hash exactly when libconvnet_hip.so is a build of this tree)
            'step_issue': step_issue,
            'build': {'binary_src_hash': ca._lib.build_hash(), 'tree_src_hash': ca._lib.source_hash(),
                      'binary_is_this_tree': ca._lib.build_hash() == ca._lib.source_hash()},
            'gpu_active_s': round(gpu_active, 2),
        }
        # measured HBM traffic of the whole step (all kernels, latest committed PMC passes: static, like
        # roofline.traffic) over this run's step time: how close the step as a whole runs to the memory system
        try:
            cands = sorted((f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('pmc_traffic.json')),
                               key=_pmc_file_order)
            if LIVE_PMC.get('pm') is not None:
                pm, cands = LIVE_PMC['pm'], ['live: rocprofv3 --pmc passes run by this command']
            else:
                pm = json.load(open(os.path.join(ROOT, 'profiles', cands[-1])))
                cands[-1] = 'static, ' + cands[-1]
            if 'hbm_bytes_per_step' in pm and args.depth == 50 and args.dtype == 'bf16' and B == 256 \
                    and not args.quantize and world == 1:
                rate = pm['hbm_bytes_per_step'] / (elapsed / args.steps) / 1e9
                out['hbm_measured_whole_step'] = {
                    'traffic_gb_per_step': round(pm['hbm_bytes_per_step'] / 1e9, 1), 'rate_gbs': round(rate, 0),
                    'frac_of_peak': round(rate / PEAK_HBM_GBS, 4),
                    'frac_of_streaming_rate': round(rate / 5100.0, 4),
                    'note': 'traffic: %s (all kernels of a step); streaming rate = 5100 GB/s (tools/bench_skew.py)'
                            % cands[-1]}
        except Exception:
            pass
        detail = {'kernels': kernels, 'kernels_overlapped': kernels_ovl, 'conv_layers': layers}
        if kernels:
            try:
                os.makedirs(os.path.dirname(os.path.abspath(args.detail_out)), exist_ok=True)
                with open(args.detail_out, 'w') as f:
                    json.dump(dict(out, **detail), f, indent=1)
                out['detail_file'] = os.path.relpath(os.path.abspath(args.detail_out), ROOT)
            except OSError as e:
                sys.stderr.write('bench.py: detail tables not written: %s\n' % e)
            if args.detail_stdout:
                print(json.dumps({'detail': detail}))
        line = json.dumps(out)
        # the contract line must stay readable by whatever tails stdout: compact, and the LAST line
        assert len(line) < 8192, 'contract line grew to %d bytes' % len(line)
    # The LAST line of the job's stdout: RCCL prints its version banner through C stdio ("RCCL version : ...", five lines,
    # rank 0), which a pipe buffers until the process exits - i.e. BEHIND a line Python printed earlier (seen with the
    # 1-rank RCCL run of this script).  So: every rank tears down its communicators and flushes C stdio, the other
    # ranks leave without running exit handlers, rank 0 prints the line after them and leaves the same way.
    if distributed:
        ca.comm.destroy_default()
        try:
            dist.barrier()
        except Exception:
            pass
        dist.destroy_process_group()
    _flush_c_stdio()
    if rank != 0:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    if distributed and world > 1:
        time.sleep(1.0)      # the other ranks' last bytes reach the launcher's pipe first
    print(line, flush=True)
    sys.stderr.flush()
    if distributed:
        os._exit(0)          # no exit handler (RCCL / c10d destructors) gets to write behind the line


if __name__ == '__main__':
    main()
