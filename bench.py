#!/usr/bin/env python3
"""Headline benchmark: images/sec of one TRAIN STEP of mobilenetv2-fpn, 512x512, batch 16 per GPU (BASELINE.json,
configs[1]), synthetic data, fp32, on N MI355X of one node (one process per GPU, RCCL gradient all-reduce).

    python bench.py                                   # N=1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward (conv backbone + FPN neck + 3 YOLO heads + fused decode/loss) + zero_grad + backward
(+ gradient all-reduce) + fused Adam update, i.e. the body of the reference's ``trainer.py:225-242``; the batch is
resident in HBM before the timed region.  Rank 0 prints ONE JSON line (contract in the task statement) carrying
``roofline`` (dominant kernel class, measured live with HIP events on the launch stream in a separate profiling pass)
and ``cpu_baseline`` (the CPU oracle's train step on the host cores, N=1 only).
"""
import argparse
import ctypes as C
import io
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--model', default='mobilenetv2-fpn',
                    help='BASELINE.json configs: mobilenetv2-fpn (the metric), regnetx-600m-fpn / -pan (configs 3, 4); '
                         'any cfg name of pqdet_amd.model.zoo')
    ap.add_argument('--qat', action='store_true', help='BASELINE config 5: quantisation-aware training step (use --batch 12)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=3, help='timed steps of the CPU oracle leg (SURVEY 8d: >= 3)')
    return ap.parse_args(argv)


def launch_ranks(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes (this parent has not touched the GPU
    and never does), one per GPU, and relay them.  Rank 0's stdout carries the JSON line.  Mirrors what
    ``python -m torch.distributed.run --nproc-per-node N`` sets up (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    from pqdet_amd.launch import wait_ranks        # pure-Python module: importing it does not touch the GPU
    return wait_ranks(procs)


def resolve_world(args):
    """(world, rank, local) from the launcher's environment; --gpus must agree with it.  Exits non-zero on a mismatch:
    a silent one-GPU run printed as an N-GPU line would be an invalid measurement."""
    if 'WORLD_SIZE' not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args))
        return 1, 0, 0
    world = int(os.environ['WORLD_SIZE'])
    if world != args.gpus:
        print('[bench] --gpus %d does not match WORLD_SIZE=%d of the launcher: refusing to run' % (args.gpus, world),
              file=sys.stderr, flush=True)
        sys.exit(2)
    return world, int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))


if __name__ == '__main__':
    # decided BEFORE torch / the HIP library are imported: the parent of a self-launched multi-rank run stays GPU-free
    _ARGS = parse_args()
    _WORLD = resolve_world(_ARGS)

import torch  # noqa: E402

from pqdet_amd import abi, optim  # noqa: E402
from pqdet_amd.dataset import labels  # noqa: E402
from pqdet_amd.model import zoo  # noqa: E402
from pqdet_amd.model.interpreter import DetectionModel  # noqa: E402

PEAK_FP32_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 MFMA == fp32 vector peak
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E spec

KERNEL_CLASS = {
    abi.OP_CONV1X1_FWD: ('conv1x1_fwd (gemm_nt, fp32 MFMA)', 'mfma'),
    abi.OP_CONV1X1_DGRAD: ('conv1x1_dgrad (gemm_nt, fp32 MFMA)', 'mfma'),
    abi.OP_CONV1X1_WGRAD: ('conv1x1_wgrad (gemm_tn, fp32 MFMA)', 'mfma'),
    abi.OP_CONV1X1_BWD: ('conv1x1_bwd narrow layers (fused dgrad+wgrad, one pass over dy)', 'hbm'),
    abi.OP_DW_FWD: ('dw3x3_fwd', 'hbm'), abi.OP_DW_DGRAD: ('dw3x3_dgrad', 'hbm'), abi.OP_DW_WGRAD: ('dw3x3_wgrad', 'hbm'),
    abi.OP_DW_BWD: ('dw3x3_bwd (both gradients, one pass: walking kernel)', 'hbm'),
    abi.OP_GCONV_FWD: ('gconv3x3_fwd (implicit GEMM, fp32 MFMA)', 'mfma'),
    abi.OP_GCONV_DGRAD: ('gconv3x3_dgrad (implicit GEMM, fp32 MFMA)', 'mfma'),
    abi.OP_GCONV_WGRAD: ('gconv3x3_wgrad (implicit GEMM, fp32 MFMA)', 'mfma'),
    abi.OP_STEM_FWD: ('stem3x3_fwd', 'hbm'), abi.OP_STEM_WGRAD: ('stem3x3_wgrad', 'hbm'),
    abi.OP_BN_FWD_FIN: ('bn_fwd_finalize', 'hbm'), abi.OP_BN_BWD_FIN: ('bn_bwd_finalize', 'hbm'),
    abi.OP_COPY_FWD: ('copy_fwd (shortcut/route/upsample)', 'hbm'), abi.OP_ROUTE_BWD: ('route_bwd', 'hbm'),
    abi.OP_LOSS_FWD: ('yolo_loss_fwd', 'hbm'), abi.OP_LOSS_BWD: ('yolo_loss_bwd', 'hbm'), abi.OP_FILL: ('fill', 'hbm'),
    abi.OP_TRANSPOSE: ('weight transposes (one launch)', 'hbm'),
    abi.OP_BN_EVAL: ('bn vectors from running statistics (one launch)', 'hbm'),
    abi.OP_QAT_WPREP: ('qat weight scale+observe+quantise (one launch)', 'hbm'),
    abi.OP_QACT_FWD: ('qat activation observe+quantise', 'hbm'), abi.OP_QACT_BWD: ('qat activation backward', 'hbm'),
    abi.OP_XSTAT_FWD: ('xstat_fwd (expanded-tensor BN statistics from input moments)', 'hbm'),
    abi.OP_XDW_FWD: ('xdw_fwd (fused expand 1x1 + BN + ReLU6 + dw3x3)', 'hbm'),
    abi.OP_XDW_BWD: ('xdw_bwd (fused dw3x3 + expand 1x1 backward, recompute)', 'hbm'),
    abi.OP_REDUCE_BATCH: ('weight-gradient partial sums (batched reduce)', 'hbm'),
}


def op_work(op):
    """Algorithmic (flops, bytes) of one op: SURVEY 8(d) counting rule -- 2*MACs; 4*(in + out elements) + weights."""
    c = op.code
    if c in (abi.OP_CONV1X1_FWD,):
        a = op.u.c1f
        return 2.0 * a.M * a.I * a.O, 4.0 * (a.M * a.I + a.M * a.O + a.I * a.O)
    if c in (abi.OP_CONV1X1_DGRAD, abi.OP_CONV1X1_WGRAD):
        a = op.u.c1d if c == abi.OP_CONV1X1_DGRAD else op.u.c1w
        return 2.0 * a.M * a.I * a.O, 4.0 * (a.M * a.I + a.M * a.O + a.I * a.O)
    if c == abi.OP_CONV1X1_BWD:      # both GEMMs; dy and the layer input are counted once each, plus the input gradient
        a = op.u.c1b
        return 2.0 * 2.0 * a.M * a.I * a.O, 4.0 * (2 * a.M * a.I + a.M * a.O + 2 * a.I * a.O)
    if c in (abi.OP_DW_FWD, abi.OP_DW_DGRAD, abi.OP_DW_WGRAD):
        a = {abi.OP_DW_FWD: op.u.dwf, abi.OP_DW_DGRAD: op.u.dwd, abi.OP_DW_WGRAD: op.u.dww}[c]
        n_in, n_out = a.B * a.IH * a.IW * a.C, a.B * a.OH * a.OW * a.C
        return 2.0 * 9 * n_out, 4.0 * (n_in + n_out + 9 * a.C)
    if c == abi.OP_DW_BWD:
        a = op.u.dwb
        n_in, n_out = a.B * a.IH * a.IW * a.C, a.B * a.OH * a.OW * a.C
        return 2.0 * 2.0 * 9 * n_out, 2.0 * 4.0 * (n_in + n_out + 9 * a.C)
    if c in (abi.OP_GCONV_FWD, abi.OP_GCONV_DGRAD, abi.OP_GCONV_WGRAD):
        a = {abi.OP_GCONV_FWD: op.u.gcf, abi.OP_GCONV_DGRAD: op.u.gcd, abi.OP_GCONV_WGRAD: op.u.gcw}[c]
        n_in, n_out = a.B * a.IH * a.IW * a.C, a.B * a.OH * a.OW * a.C
        return 2.0 * 9 * a.gw * n_out, 4.0 * (n_in + n_out + 9 * a.gw * a.C)
    if c in (abi.OP_STEM_FWD, abi.OP_STEM_WGRAD):
        a = op.u.stf if c == abi.OP_STEM_FWD else op.u.stw
        n_in, n_out = a.B * a.CI * a.IH * a.IW, a.B * a.OH * a.OW * a.O
        return 2.0 * 27 * n_out, 4.0 * (n_in + n_out)
    if c == abi.OP_COPY_FWD:
        a = op.u.cp
        n = a.B * a.H * a.W * a.C
        return float(n), 4.0 * n * (3 if a.b.x else 2)
    if c == abi.OP_ROUTE_BWD:
        a = op.u.rb
        n = a.B * a.H * a.W * a.C
        return float(n), 4.0 * n * 2
    # fused inverted-residual ops: the counting rule of the layers they replace (expand 1x1 + depthwise 3x3)
    if c == abi.OP_XSTAT_FWD:
        a = op.u.xsf
        return 2.0 * a.M * a.I * a.I, 4.0 * a.M * a.I
    if c == abi.OP_XDW_FWD:
        a = op.u.xdf
        m_in, n_out = a.B * a.IH * a.IW, a.B * a.OH * a.OW * a.E
        return 2.0 * m_in * a.I * a.E + 2.0 * 9 * n_out, 4.0 * (m_in * a.I + m_in * a.E + a.I * a.E) + 4.0 * (m_in * a.E + n_out + 9 * a.E)
    if c == abi.OP_XDW_BWD:
        a = op.u.xdb
        m_in, n_out = a.B * a.IH * a.IW, a.B * a.OH * a.OW * a.E
        return (2.0 * 2.0 * m_in * a.I * a.E + 2.0 * 2.0 * 9 * n_out,
                4.0 * (2 * m_in * a.I + m_in * a.E + 2 * a.I * a.E) + 2.0 * 4.0 * (m_in * a.E + n_out + 9 * a.E))
    if c in (abi.OP_LOSS_FWD, abi.OP_LOSS_BWD):
        a = op.u.loss
        n = a.B * a.H * a.W * a.A * (5 + a.C)
        return 30.0 * n, 4.0 * n * 3
    return 0.0, 0.0


def profile_kernels(plan, x, tgt, side=None, passes=3):
    """Per-kernel-class device time of one step: HIP events around every op on the stream it is launched on, replayed
    exactly as in the timed steps (weight-gradient kernels overlap the data-gradient chain on the side stream), so the
    averages are directly comparable with a rocprofv3 kernel trace of this command (profiles/r2_kernel_stats.csv)."""
    l = abi.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sd = C.c_void_p(side.cuda_stream) if side is not None else None
    plan._patch(x, tgt)
    acc = {}
    for arr, n in ((plan.fwd_arr, len(plan.fwd)), (plan.bwd_arr, len(plan.bwd))):
        ms = (C.c_float * n)()
        tot = [0.0] * n
        for _ in range(passes):
            abi.check(l.pq_run_ops2_timed(arr, n, st, sd, ms), 'profiling pass')
            for k in range(n):
                tot[k] += ms[k] / passes
        for k in range(n):
            if arr[k].code == abi.OP_FILL and arr[k].u.fill.n == 0:
                continue                 # a place-holder op of the plan (no launch): its event pair is not device time
            name, bound = KERNEL_CLASS.get(arr[k].code, ('other', 'hbm'))
            fl, by = op_work(arr[k])
            e = acc.setdefault(name, {'ms': 0.0, 'flops': 0.0, 'bytes': 0.0, 'launches': 0, 'bound': bound})
            e['ms'] += tot[k]; e['flops'] += fl; e['bytes'] += by; e['launches'] += 1
    return acc


# kernel class -> (regex of the class's main kernel, regex of helper kernels launched by the same op) over the kernel
# names of profiles/r1_pmc_summary.csv (gemm_nt template args: BM,BN,WM,WN,AMODE,BKN,EMODE,..)
PMC_PATTERN = {
    # gemm_nt template args: BM,BN,WM,WN,AMODE,BKN,EMODE,..;  gemm2_nt: BM,BN,AMODE,EMODE,S  (EMODE 2 = gradient sink)
    'conv1x1_dgrad (gemm_nt, fp32 MFMA)': (r'gemm_nt_kernel<\d+, \d+, \d+, \d+, \d, \w+, 2, |gemm2_nt_kernel<\d+, \d+, \d, 2, ', None),
    'conv1x1_fwd (gemm_nt, fp32 MFMA)': (r'gemm_nt_kernel<\d+, \d+, \d+, \d+, \d, \w+, [01], |gemm2_nt_kernel<\d+, \d+, \d, [01], ', None),
    'conv1x1_wgrad (gemm_tn, fp32 MFMA)': (r'gemm_tn_kernel<|gemm2_tn_kernel<', r'pq::slab_reduce|pq::reduce_batch_kernel'),
    'conv1x1_bwd narrow layers (fused dgrad+wgrad, one pass over dy)': (r'pw_bwd_kernel<', r'pw_slab_reduce_kernel'),
    'gconv3x3_fwd (implicit GEMM, fp32 MFMA)': (r'gconv_kernel<[01], ', None),
    'gconv3x3_dgrad (implicit GEMM, fp32 MFMA)': (r'gconv_kernel<2, |gconv_dgrad_s2_kernel', None),
    'gconv3x3_wgrad (implicit GEMM, fp32 MFMA)': (r'gconv_wgrad_kernel<', r'gconv_wgrad_reduce|gconv_slab'),
}
# per (model, QAT plan): the PMC summary collected with scripts/profile_round.sh for exactly that plan (MODEL / BENCH_ARGS)
PMC_FILES = {('mobilenetv2-fpn', False): 'r5_pmc_summary.csv', ('regnetx-600m-fpn', False): 'r5_pmc_summary_regnetx-600m-fpn.csv',
             ('regnetx-600m-pan', False): 'r5_pmc_summary_regnetx-600m-pan.csv', ('mobilenetv2-fpn', True): 'r5_pmc_summary_qat.csv'}


def measured_traffic(kernel_class, model_name, qat=False):
    """HBM bytes per launch of a kernel class from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs of this command, gfx950 read correction applied -- scripts/pmc_summary.py), averaged over the launches of
    the class's kernels (both kernel generations) with the helper launches of the same op added in.  None if unavailable."""
    import csv
    import re
    fname = PMC_FILES.get((model_name, bool(qat)))
    main, extra = PMC_PATTERN.get(kernel_class, (None, None))
    if fname is None or main is None:
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', fname)
    if not os.path.exists(path):
        return None
    calls = tot = 0.0
    for r in csv.DictReader(open(path)):
        is_main = re.search(main, r['kernel']) is not None
        if is_main or (extra and re.search(extra, r['kernel'])):
            n = float(r['calls'])
            calls += n if is_main else 0.0
            tot += n * (float(r['hbm_read_mb_per_launch(2x)']) + float(r['hbm_write_mb_per_launch'])) * 1e6
    return round(tot / calls) if calls else None


def log(*a):
    print('[bench]', *a, file=sys.stderr, flush=True)


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(batch, size, steps=2, model='mobilenetv2-fpn', qat=False):
    """The CPU oracle (oracle/, PyTorch-CPU restatement pinned against the reference's golden vectors) timed on the
    host cores on the same workload shape.  Reported baseline only -- never part of the measured path."""
    from oracle.net import OracleNet
    threads = usable_cores()
    torch.set_num_threads(threads)
    log('cpu baseline on %d threads (os.cpu_count=%s)' % (threads, os.cpu_count()))
    if qat:                                        # config 5: the oracle's quantisation-aware net (observers + fake-quant on)
        from oracle.qat import OracleQATNet
        net = OracleQATNet(io.StringIO(zoo.cfg_text(model)))
    else:
        net = OracleNet(io.StringIO(zoo.cfg_text(model)))
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=4e-4)
    x = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(0))
    tgt = labels.synthetic_targets(batch, size, 20, seed=1)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        net(x, tgt)['loss'].mean().backward()
        opt.step()
        log('cpu step %d: %.2f s' % (i, time.perf_counter() - t0))
        if i:
            times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    return {'value': batch / dt, 'unit': 'images/sec', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d timed train steps (after 1 warm-up) of %s%s %dx%d batch %d on the CPU oracle, '
                      '%.2f s/step' % (steps, model, ' (QAT)' if qat else '', size, size, batch, dt)}


def main(args, world, rank, local):
    if args.model not in zoo.available():
        sys.exit('[bench] unknown model %r (have: %s)' % (args.model, ', '.join(zoo.available())))
    dist = None
    rehearsal = os.environ.get('PQ_BENCH_REHEARSAL') == '1'     # all ranks on cuda:0 over gloo: exercises this file's
    if rehearsal:                                               # multi-rank flow on a one-GPU box (never a measurement)
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    if os.environ.get('PQ_MAIN_PRIO') == '1':                   # A/B only: the step on a high-priority stream (weight gradients stay normal)
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    torch.manual_seed(0)
    model = DetectionModel(io.StringIO(zoo.cfg_text(args.model)), quant=args.qat).to(dev)
    model.train()
    if args.qat:
        from pqdet_amd import tools
        tools.prepare_qat(tools.fuse_model(model), backend='qnnpack')
    model.nan_check = 'deferred'
    runner = model
    if world > 1:
        from pqdet_amd.parallel import DistributedModel
        runner = DistributedModel(model)
    opt = optim.Adam(model.parameters(), lr=4e-4, weight_decay=0.0)

    B, S = args.batch, args.size
    x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(rank)).to(dev)
    tgt = tuple(t.to(dev) for t in labels.synthetic_targets(B, S, 20, seed=1 + rank))

    def step():
        opt.zero_grad()
        out = runner(x, tgt)
        out['loss'].mean().backward()
        opt.step()
        return out

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    log('rank %d/%d: model + batch on %s, warming up' % (rank, world, dev))
    for _ in range(args.warmup):
        step()
    fence()
    log('timing %d steps' % args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    model.check_nan(wait=True)
    loss_value = out['loss'].item()

    log('timed region done: %.3f s' % dt)
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        plan = list(model._plans.values())[0]
        prof = profile_kernels(plan, x, tgt, side=model._side_stream())
        dom_name, dom = max(prof.items(), key=lambda kv: kv[1]['ms'])
        if dom['bound'] == 'mfma':
            achieved, peak, unit = dom['flops'] / (dom['ms'] * 1e-3) / 1e12, PEAK_FP32_TFLOPS, 'TFLOP/s'
        else:
            achieved, peak, unit = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9, PEAK_HBM_GBS, 'GB/s'
        total_ms = sum(v['ms'] for v in prof.values())
        roofline = {'bound': dom['bound'], 'achieved': round(achieved, 2), 'peak': peak, 'unit': unit,
                    'frac': round(achieved / peak, 4), 'traffic': measured_traffic(dom_name, args.model, getattr(args, 'qat', False)),
                    'traffic_unit': 'HBM bytes per launch (profiles/%s)' % PMC_FILES.get((args.model, bool(getattr(args, 'qat', False))), 'none'),
                    'algorithmic_per_launch': round((dom['flops'] if dom['bound'] == 'mfma' else dom['bytes']) / dom['launches']),
                    'algorithmic_bytes_per_launch': round(dom['bytes'] / dom['launches']), 'kernel': dom_name,
                    'kernel_avg_us': round(dom['ms'] * 1e3 / dom['launches'], 2),
                    'kernel_ms_per_step': round(dom['ms'], 3), 'launches_per_step': dom['launches'],
                    'device_ms_per_step_all_kernels': round(total_ms, 3),
                    'breakdown_ms': {k: round(v['ms'], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])}}
        line = {
            'metric': 'images/sec %strain step, %s %dx%d bs%d/GPU' % ('QAT ' if args.qat else '', args.model, S, S, B),
            'value': round(value, 2),
            'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s.cfg train step (fwd + YOLO loss + bwd + Adam), %dx%d, batch %d per GPU, '
                                   'C=20, random-init weights' % (args.model, S, S, B),
                       'global_batch': world * B, 'parallelism': 'dp%d' % world, 'final_loss': round(loss_value, 4)},
            'roofline': roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(B, S, steps=max(1, args.cpu_steps), model=args.model, qat=args.qat)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main(_ARGS, *_WORLD)
