"""Run a driver written against the REFERENCE's module names on this package, one process per GPU.

    python -m pqdet_amd.launch [--nproc N] [--backend nccl|gloo] script.py [script args ...]

The reference is single-process ``nn.DataParallel`` (``tools.py:215-216``, ``trainer.py:225-242``); here every GPU gets its
own process (RCCL underneath ``torch.distributed``'s ``nccl`` backend).  For each rank the launcher

1. pins the rank's device (``LOCAL_RANK``) and initialises the process group *before* the driver runs, so that
   ``tools.build_model(..., dataparallel=True)`` returns the gradient-exchanging ``DistributedModel``;
2. installs ``tools``, ``model`` / ``model.interpreter`` / ``model.parser`` / ``model.loss``, ``dataset`` and
   ``eval`` / ``eval.evaluator`` in ``sys.modules`` as aliases of the ``pqdet_amd`` modules of the same role, so a driver
   that says ``import tools; from model.interpreter import DetectionModel`` runs with zero edits;
3. executes the script as ``__main__``.

With ``--nproc N`` (and no ``WORLD_SIZE`` in the environment) it first starts N fresh copies of itself with ``RANK`` /
``LOCAL_RANK`` / ``WORLD_SIZE`` / ``MASTER_*`` set -- the parent never touches the GPU; under ``torch.distributed.run``
those variables already exist and the launcher only does steps 1-3.
"""
import argparse
import importlib
import os
import runpy
import socket
import subprocess
import sys

ALIASES = {
    'tools': 'pqdet_amd.tools',
    'model': 'pqdet_amd.model',
    'model.interpreter': 'pqdet_amd.model.interpreter',
    'model.parser': 'pqdet_amd.model.parser',
    'model.loss': 'pqdet_amd.model.loss',
    'dataset': 'pqdet_amd.dataset',
    'eval': 'pqdet_amd.eval',
    'eval.evaluator': 'pqdet_amd.eval.evaluator',
}


def install_aliases():
    """Make the reference's import names resolve to this package (idempotent)."""
    for alias, real in ALIASES.items():
        sys.modules[alias] = importlib.import_module(real)


def _spawn(nproc: int, argv) -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, '-m', 'pqdet_amd.launch'] + argv, env=env))
    return wait_ranks(procs)


def wait_ranks(procs, poll_s: float = 0.2, grace_s: float = 10.0) -> int:
    """Wait for the rank processes.  The first rank that exits non-zero ends the job: its peers would block for ever in a
    collective it never joins, so they are terminated (killed after ``grace_s``) and that exit code is returned."""
    import time
    live = list(procs)
    rc = 0
    while live and rc == 0:
        for p in list(live):
            r = p.poll()
            if r is not None:
                live.remove(p)
                rc = rc or r
        if live and rc == 0:
            time.sleep(poll_s)
    if live:
        for p in live:
            p.terminate()
        deadline = time.monotonic() + grace_s
        for p in live:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser(prog='python -m pqdet_amd.launch', description=__doc__.split('\n\n')[0])
    ap.add_argument('--nproc', type=int, default=0, help='start this many rank processes (default: use the environment)')
    ap.add_argument('--backend', default=os.environ.get('PQ_LAUNCH_BACKEND', 'nccl'),
                    help="torch.distributed backend; 'gloo' rehearses on CPU tensors / one GPU")
    ap.add_argument('script')
    ap.add_argument('args', nargs=argparse.REMAINDER)
    ns = ap.parse_args(argv)
    if ns.nproc > 1 and 'WORLD_SIZE' not in os.environ:
        return _spawn(ns.nproc, ['--backend', ns.backend, ns.script] + ns.args)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    import torch
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if ns.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', torch.cuda.current_device()))
        else:
            dist.init_process_group(ns.backend)
    install_aliases()
    sys.argv = [ns.script] + ns.args
    sys.path.insert(0, os.path.dirname(os.path.abspath(ns.script)))
    # a driver that raises (or exits non-zero) must NOT enter the closing barrier: its peers are inside other collectives and
    # would never match it; this rank exits non-zero instead and the parent ends the job (wait_ranks)
    # (a driver that ends with sys.exit(0) / sys.exit(main()) returning 0 is a clean end: it falls through to the barrier like a
    # script that simply runs off its last line)
    try:
        runpy.run_path(ns.script, run_name='__main__')
    except SystemExit as e:
        if e.code not in (None, 0):
            raise
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
