"""A rank that fails must end the whole launch: no closing barrier on the failing rank, the parent terminates
the peers that are blocked in a collective and returns the failing rank's exit code."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = '''
import os, sys
import torch.distributed as dist
if int(os.environ['RANK']) == 1:
    raise RuntimeError('rank 1 fails before its first collective')
dist.barrier()              # rank 0 would wait here for ever
print('unreachable')
'''


def test_failing_rank_ends_the_launch(tmp_path):
    script = tmp_path / 'driver.py'
    script.write_text(DRIVER)
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    env['CUDA_VISIBLE_DEVICES'] = env['HIP_VISIBLE_DEVICES'] = ''          # CPU rehearsal (gloo)
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, '-m', 'pqdet_amd.launch', '--nproc', '2', '--backend', 'gloo', str(script)], env=env,
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0
    assert 'rank 1 fails' in p.stderr and 'unreachable' not in p.stdout
    assert time.monotonic() - t0 < 240


def test_wait_ranks_returns_first_failure_and_terminates_the_rest():
    from pqdet_amd.launch import wait_ranks
    sleeper = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])
    failer = subprocess.Popen([sys.executable, '-c', 'import sys; sys.exit(7)'])
    t0 = time.monotonic()
    assert wait_ranks([sleeper, failer], grace_s=5.0) == 7
    assert sleeper.poll() is not None and time.monotonic() - t0 < 60
    ok = [subprocess.Popen([sys.executable, '-c', 'pass']) for _ in range(2)]
    assert wait_ranks(ok) == 0


CLEAN_EXIT_DRIVER = '''
import os, sys, time
import torch.distributed as dist
dist.barrier()
if int(os.environ['RANK']) == 0:
    sys.exit(0)              # one rank ends through SystemExit(0), the other runs off its last line a little later
time.sleep(1.0)
print('rank 1 done')
'''


def test_clean_sys_exit_still_reaches_the_closing_barrier(tmp_path):
    """sys.exit(0) in the driver is a clean end -- the rank must still enter the closing barrier and destroy its
    process group, or a peer that ends later waits for it until the rendezvous times out."""
    script = tmp_path / 'driver.py'
    script.write_text(CLEAN_EXIT_DRIVER)
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    env['CUDA_VISIBLE_DEVICES'] = env['HIP_VISIBLE_DEVICES'] = ''
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, '-m', 'pqdet_amd.launch', '--nproc', '2', '--backend', 'gloo', str(script)], env=env,
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    assert 'rank 1 done' in p.stdout
    assert time.monotonic() - t0 < 120
