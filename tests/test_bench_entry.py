"""bench.py's multi-rank entry: ``--gpus N`` must either run N ranks or refuse -- never time one GPU
and print it as N.  The world-size decision is taken before torch / the HIP library are imported, so the refusal path
runs without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def test_gpus_flag_must_match_launcher_world_size():
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, BENCH, '--gpus', '4'], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert 'does not match WORLD_SIZE' in p.stderr
    assert p.stdout.strip() == ''          # no JSON line for an invalid configuration


def test_defaults_follow_survey_8d():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module('bench')
    a = bench.parse_args([])
    assert a.gpus == 1 and a.steps >= 100 and a.warmup >= 20 and a.cpu_steps >= 3


@pytest.mark.gpu
def test_self_launched_two_rank_rehearsal_reports_two_gpus():
    """No launcher, ``--gpus 2``: bench.py starts its two rank processes itself (the parent never touches the GPU).
    PQ_BENCH_REHEARSAL=1 puts both ranks on cuda:0 over gloo -- the one-GPU box's stand-in for RCCL; not a measurement."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['PQ_BENCH_REHEARSAL'] = '1'
    p = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2', '--size', '128',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 4 and line['config']['parallelism'] == 'dp2'
    assert line['steps'] == 2 and line['warmup'] == 1
