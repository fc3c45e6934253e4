import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, GOLDEN_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)


def _usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (bench.py::usable_cores)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the CPU oracle (PyTorch-CPU) sizes its thread pool by the HOST's core count; a GPU box hands a job 16 of 256 cores through a
    # cgroup quota, and 128 threads on 16 cores run every oracle pass at half speed (measured: 252 -> 116 ms for a depthwise pass)
    import torch
    torch.set_num_threads(min(torch.get_num_threads(), _usable_cores()))


@pytest.fixture(scope='session')
def golden():
    return np.load(os.path.join(GOLDEN_DIR, 'golden.npz'))


@pytest.fixture(scope='session')
def g1():
    with open(os.path.join(GOLDEN_DIR, 'g1_parser.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def g5_names():
    with open(os.path.join(GOLDEN_DIR, 'g5_param_names.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def golden_extra():
    return np.load(os.path.join(GOLDEN_DIR, 'golden_extra.npz'))


@pytest.fixture(scope='session')
def golden_r2():
    return np.load(os.path.join(GOLDEN_DIR, 'golden_r2.npz'))


@pytest.fixture(scope='session')
def r2_names():
    with open(os.path.join(GOLDEN_DIR, 'g_r2_names.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def g12():
    with open(os.path.join(GOLDEN_DIR, 'g12_checkpoints.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def golden_r3():
    return np.load(os.path.join(GOLDEN_DIR, 'golden_r3.npz'))


@pytest.fixture(scope='session')
def r3_names():
    with open(os.path.join(GOLDEN_DIR, 'g_r3_names.json')) as f:
        return json.load(f)


# ---- launch log: which kernel instantiations do the KERNEL-LEVEL value tests launch?  (tests/test_zz_instantiations_gpu.py compares
# that set with what the plans of the BASELINE nets launch.)  Recording is on only while a test of one of these modules runs.
KERNEL_TEST_MODULES = {'test_kernels_gpu', 'test_irblock_gpu', 'test_reduce_batch_gpu', 'test_gemm_variants_gpu', 'test_bn_handover_gpu', 'test_qat_gpu'}
LAUNCHES_COVERED = set()


def dump_launch_log():
    import ctypes as C
    from pqdet_amd import abi
    l = abi.lib()
    need = l.pq_launch_log_dump(None, 0)
    buf = C.create_string_buffer(int(need) + 16)
    l.pq_launch_log_dump(buf, need + 16)
    return {e for e in buf.value.decode().split('\n') if e}


@pytest.fixture(autouse=True)
def _record_kernel_test_launches(request):
    mod = request.module.__name__.split('.')[-1]
    if mod not in KERNEL_TEST_MODULES or request.node.get_closest_marker('gpu') is None:
        yield
        return
    from pqdet_amd import abi
    abi.lib().pq_launch_log(1)
    yield
    LAUNCHES_COVERED.update(dump_launch_log())
    abi.lib().pq_launch_log(0)
