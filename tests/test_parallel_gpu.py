"""Two data-parallel ranks on the one GPU of the test box (gloo transport): DistributedModel's parameter broadcast, the
bucketed gradient exchange interleaved with the two-stream backward, and the fused Adam step keep replicas identical and
produce exactly the mean of the per-rank gradients."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('model_name', ['mobilenetv2-fpn', 'regnetx-600m-fpn'])     # folded linear shortcuts / aliased ReLU shortcuts + grouped 3x3
def test_two_rank_step_on_one_gpu(model_name):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), PQ_TEST_MODEL=model_name)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'ddp_gpu_worker.py')], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (rank, out[-3000:])
        assert 'rank %d ok' % rank in out
