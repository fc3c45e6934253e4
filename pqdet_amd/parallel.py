"""Data-parallel training, one process per GPU, gradients exchanged with RCCL over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` (``tools.py:215-216``; per step: replicate parameters, scatter
the batch, gather losses, reduce gradients to device 0).  Here every rank owns a full replica and its own slice of the
global batch (BatchNorm statistics stay per-rank, exactly like DataParallel's per-replica BN); parameters are broadcast
once; each step the flat gradient buffer is averaged with a few large all-reduces.  Because gradients are produced in
reverse layer order and parameters are laid out in layer order, every all-reduce covers one contiguous tail range of the
flat buffer and is issued on a side HIP stream as soon as the backward has passed the first layer of the range, so the
exchange overlaps the rest of the backward.  xGMI is point-to-point (ring all-reduce is per-link bound), so buckets are
kept large (default 8 MiB => 4 all-reduces for the 27 MB of mobilenetv2-fpn gradients).

Averaging equals DataParallel's ``loss.mean()`` over equally sized replicas (reference ``trainer.py:233``).
"""
from typing import List, Tuple

import torch
import torch.distributed as dist
from torch import nn


class GradReducer:
    """Averages contiguous ranges of a flat gradient buffer across ranks, asynchronously on a side stream."""

    def __init__(self, bucket_bytes: int = 8 << 20, process_group=None):
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.stream = None
        self.handles = []

    def pick_marks(self, segments: List[Tuple[int, int]], total: int) -> List[Tuple[int, int]]:
        """segments = [(ops_done, flat_lo)] in backward order (flat_lo decreasing).  Choose the marks at which a
        gradient range of at least one bucket has become final."""
        marks, hi = [], total
        for op_end, lo in segments:
            if hi - lo >= self.bucket_elems:
                marks.append((op_end, lo))
                hi = lo
        return marks

    def reduce_range(self, flat_grad: torch.Tensor, lo: int, hi: int, also_wait=None):
        """also_wait: a second compute stream whose work (weight-gradient kernels) also writes this range; the communication
        stream waits for it too -- the compute streams themselves are never made to wait for each other here."""
        if hi <= lo or self.world == 1:
            return
        chunk = flat_grad[lo:hi]
        if flat_grad.is_cuda:
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=flat_grad.device)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            self.stream.wait_event(ready)
            if also_wait is not None:
                self.stream.wait_stream(also_wait)
            with torch.cuda.stream(self.stream):
                if dist.get_backend(self.group) == 'nccl':          # RCCL: averaging all-reduce, asynchronous on the stream
                    dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group)
                else:                                               # gloo on device tensors (single-GPU rehearsal): SUM, scale
                    dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
                    chunk.div_(self.world)
        else:   # gloo (CPU rehearsal): no AVG -> SUM then scale
            h = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.handles.append((h, chunk))

    def finish(self):
        for h, chunk in self.handles:
            h.wait()
            chunk.div_(self.world)
        self.handles.clear()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


class DistributedModel(nn.Module):
    """``module`` replica on this rank + gradient averaging hooked into the compiled backward."""

    def __init__(self, module: nn.Module, bucket_bytes: int = 8 << 20, process_group=None):
        super().__init__()
        self.module = module
        self.reducer = GradReducer(bucket_bytes, process_group)
        module._reducer = self.reducer if self.reducer.world > 1 else None
        self._synced = False

    def _sync_parameters(self):
        flat, _ = self.module.flat_buffers()
        dist.broadcast(flat, src=0, group=self.reducer.group)
        for b in self.module.buffers():
            dist.broadcast(b, src=0, group=self.reducer.group)
        self._synced = True

    def forward(self, *args, **kwargs):
        if not self._synced and self.reducer.world > 1:
            self._sync_parameters()
        return self.module(*args, **kwargs)
