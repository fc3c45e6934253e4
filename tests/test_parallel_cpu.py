"""world_size-2 rehearsal (gloo, CPU) of the gradient exchange used by the N>1 bench path."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pqdet_amd.parallel import GradReducer
    n = 1000
    red = GradReducer(bucket_bytes=4 * 300)
    flat = torch.zeros(n)
    # backward produces gradients tail-first: segments = (ops done, flat_lo)
    segments = [(10, 900), (20, 650), (30, 640), (40, 300), (50, 100), (60, 0)]
    marks = red.pick_marks(segments, n)
    hi = n
    produced = []
    for op_end, lo in marks:
        flat[lo:hi] = torch.arange(lo, hi, dtype=torch.float32) * (rank + 1)     # "kernels" fill this range
        red.reduce_range(flat, lo, hi)
        produced.append((lo, hi))
        hi = lo
    if hi > 0:
        flat[0:hi] = torch.arange(0, hi, dtype=torch.float32) * (rank + 1)
        red.reduce_range(flat, 0, hi)
    red.finish()
    want = torch.arange(n, dtype=torch.float32) * (sum(r + 1 for r in range(world)) / world)
    ok = torch.allclose(flat, want)
    big_enough = all(h - l >= 300 for l, h in produced)
    out.put((rank, bool(ok), bool(big_enough), len(marks)))
    dist.destroy_process_group()


def test_bucketed_average_two_ranks():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(30)
    assert all(ok and big for _, ok, big, _ in res), res
    assert res[0][3] >= 2
