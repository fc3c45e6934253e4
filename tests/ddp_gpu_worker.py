"""Worker of tests/test_parallel_gpu.py: one rank of a 2-process data-parallel step sharing cuda:0 (gloo transport, so the
rehearsal needs only one GPU; on a multi-GPU node the same code path runs over RCCL)."""
import io
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from pqdet_amd import optim  # noqa: E402
from pqdet_amd.dataset import labels  # noqa: E402
from pqdet_amd.model import zoo  # noqa: E402
from pqdet_amd.model.interpreter import DetectionModel  # noqa: E402
from pqdet_amd.parallel import DistributedModel  # noqa: E402
from seedfill import seeded_fill, seeded_images  # noqa: E402


def fresh(seed):
    net = DetectionModel(io.StringIO(zoo.cfg_text(os.environ.get('PQ_TEST_MODEL', 'mobilenetv2-fpn'))))
    seeded_fill(net, seed)
    return net.cuda().train()


def batch(rank, b=2, s=128):
    x = seeded_images(b, s, s, 10 + rank).cuda()
    tgt = tuple(t.cuda() for t in labels.synthetic_targets(b, s, 20, seed=20 + rank))
    return x, tgt


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % os.environ['MASTER_PORT'], rank=rank, world_size=world)
    torch.cuda.set_device(0)
    # ranks start from different weights: the wrapper must broadcast rank 0's
    net = fresh(seed=1 + rank)
    dm = DistributedModel(net, bucket_bytes=1 << 20)
    opt = optim.Adam(net.parameters(), lr=1e-3)
    x, tgt = batch(rank)
    opt.zero_grad()
    out = dm(x, tgt)
    out['loss'].mean().backward()
    torch.cuda.synchronize()
    flat, grad = net.flat_buffers()
    got = grad.clone()

    # what the exchange must produce: mean of the per-rank gradients, each computed by a plain single-replica step
    per_rank = []
    for r in range(world):
        ref = fresh(seed=1)
        xr, tr = batch(r)
        ref(xr, tr)['loss'].mean().backward()
        torch.cuda.synchronize()
        per_rank.append(ref.flat_buffers()[1].clone())
    want = (per_rank[0] + per_rank[1]) / world
    err = (got - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= 1e-6 * scale, 'rank %d: averaged gradient differs: %g (scale %g)' % (rank, err, scale)
    own = (got - per_rank[rank]).abs().max().item()
    assert own > 1e-4 * scale, 'gradients were not exchanged'
    opt.step()
    torch.cuda.synchronize()
    # replicas stay in lock-step after the update
    mine = flat.clone()
    other = flat.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(mine, other), 'rank %d: parameters diverged after the step' % rank
    dist.barrier()
    dist.destroy_process_group()
    print('rank %d ok: max |avg grad - expected| = %.3g (scale %.3g)' % (rank, err, scale), flush=True)


if __name__ == '__main__':
    main()
