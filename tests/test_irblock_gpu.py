"""Fused inverted-residual block kernels (csrc/irblock.hip) against a plain PyTorch-CPU statement of the same arithmetic and
against the unfused kernels they replace (pq_conv1x1_fwd -> pq_bn_fwd_finalize -> pq_dw3x3_fwd).

mirrors: conv(1x1) -> BatchNorm -> ReLU6 -> conv(3x3, groups = C) triples of mobilenetv2-fpn.cfg (reference model/parser.py:389-410).
Tolerance: values 1e-3 abs-or-rel (BASELINE.json), in practice 1e-5 (norm-wise 2e-5 asserted by close()).
"""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from pqdet_amd import abi  # noqa: E402
from test_kernels_gpu import DEV, chan_vec, close, reduce_partial, rnd, stream  # noqa: E402


def _case(B, H, W, I, E, stride, seed=0):
    g = torch.Generator().manual_seed(1000 * B + 10 * H + W + I + E + stride + seed)
    x = rnd(B, H, W, I, g=g)
    xsc, xsh = torch.rand(I, generator=g) + 0.5, rnd(I, g=g, scale=0.3)
    we = rnd(E, I, g=g, scale=1 / math.sqrt(I))
    wd = rnd(E, 1, 3, 3, g=g, scale=0.4)
    return g, x, xsc, xsh, we, wd


def _reference(x, xsc, xsh, xact, we, msc, msh, wd, stride):
    xin = x * xsc + xsh
    if xact == 2:
        xin = xin.clamp(0, 6)
    ye = xin @ we.t()                                            # (B, H, W, E) raw expanded tensor
    h = (ye * msc + msh).clamp(0, 6)
    y = F.conv2d(h.permute(0, 3, 1, 2), wd, stride=stride, padding=1, groups=wd.shape[0])
    return ye, y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize('B,H,W,I,E,stride,xact', [
    (2, 24, 40, 24, 144, 1, 0),       # residual block input (materialised sum): no activation on the operand
    (1, 19, 21, 24, 144, 1, 0),       # odd plane, partial strips and steps
    (2, 16, 32, 16, 96, 2, 0),
    (1, 40, 72, 16, 96, 2, 0),        # more than one strip and segment at stride 2
    (2, 24, 24, 32, 192, 1, 0),
    (1, 20, 36, 32, 192, 2, 0),
    (2, 12, 20, 24, 40, 2, 0),        # E not a multiple of the 32-channel chunk
    (1, 9, 13, 16, 36, 1, 0),
    (1, 128, 128, 24, 144, 1, 0),     # benchmark-step plane sizes (one image)
    (1, 256, 256, 16, 96, 2, 0),
])
def test_xdw_fwd_matches_the_unfused_chain(B, H, W, I, E, stride, xact):
    l = abi.lib()
    g, x, xsc, xsh, we, wd = _case(B, H, W, I, E, stride)
    msc, msh = torch.rand(E, generator=g) * 0.5 + 0.25, rnd(E, g=g, scale=0.5) + 1.0
    ye, want = _reference(x, xsc, xsh, xact, we, msc, msh, wd, stride)
    OH, OW = want.shape[1:3]
    P = (E + 3) // 4 * 4
    nblk = l.pq_xdw_fwd_blocks(B, H, W, I, E, stride)
    assert nblk > 0
    ldp = (E + 15) // 16 * 16
    xd, wed, wdd = x.to(DEV), we.to(DEV), wd.to(DEV)
    vec = [chan_vec(v) for v in (xsc, xsh, msc, msh)]
    out = torch.full((B, OH, OW, P), float('nan'), device=DEV)
    yexp = torch.full((B, H, W, P), float('nan'), device=DEV)
    partial = torch.full((nblk * 2 * ldp,), float('nan'), device=DEV)
    a = abi.XdwFwd()
    a.inp = abi.Src(xd.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), I, xact)
    a.we, a.wd = wed.data_ptr(), wdd.data_ptr()
    a.mid = abi.Src(None, vec[2].data_ptr(), vec[3].data_ptr(), 0, 2)
    a.yexp, a.out, a.partial = yexp.data_ptr(), out.data_ptr(), partial.data_ptr()
    a.B, a.IH, a.IW, a.OH, a.OW, a.I, a.E, a.stride = B, H, W, OH, OW, I, E, stride
    a.ldy, a.ldo, a.ldp = P, P, ldp
    abi.check(l.pq_xdw_fwd(C.byref(a), stream()))
    torch.cuda.synchronize()
    close(out[..., :E], want, what='fused expand + depthwise output')
    close(yexp[..., :E], ye, what='raw expanded tensor')
    s1, s2 = reduce_partial(partial, nblk, ldp, E)
    close(s1.float(), want.double().sum((0, 1, 2)).float(), rt=1e-3, at=1e-2, what='sum')
    close(s2.float(), (want.double() ** 2).sum((0, 1, 2)).float(), rt=1e-3, at=1e-2, what='sum of squares')
    # without the optional expanded output: same values
    out2 = torch.full((B, OH, OW, P), float('nan'), device=DEV)
    a.yexp, a.out = None, out2.data_ptr()
    abi.check(l.pq_xdw_fwd(C.byref(a), stream()))
    torch.cuda.synchronize()
    assert torch.equal(out2[..., :E], out[..., :E])


@pytest.mark.parametrize('B,H,W,I,E,stride', [(2, 32, 32, 24, 144, 1), (2, 32, 32, 16, 96, 2), (2, 16, 48, 32, 192, 1)])
def test_xdw_fwd_with_statistics_handed_over_by_the_statistics_only_expand_pass(B, H, W, I, E, stride):
    """pq_conv1x1_fwd(out = NULL, stat_acc) -> pq_xdw_fwd(mid.stat): train-mode BatchNorm of the expanded tensor without the tensor,
    without a finalize launch; running statistics updated by the consumer's first workgroup like nn.BatchNorm2d does."""
    l = abi.lib()
    g, x, xsc, xsh, we, wd = _case(B, H, W, I, E, stride, seed=7)
    gamma, beta = torch.rand(E, generator=g) + 0.5, rnd(E, g=g, scale=0.3)
    eps, mom = 1e-5, 0.1
    M = B * H * W
    xin = x * xsc + xsh
    ye = (xin @ we.t()).view(M, E)
    mean, var = ye.double().mean(0), ye.double().var(0, unbiased=False)
    inv = 1 / torch.sqrt(var + eps)
    msc, msh = (gamma.double() * inv).float(), (beta.double() - mean * gamma.double() * inv).float()
    _, want = _reference(x, xsc, xsh, 0, we, msc, msh, wd, stride)
    OH, OW = want.shape[1:3]
    P, ldp = E, (E + 15) // 16 * 16
    xd, wed, wdd = x.to(DEV), we.to(DEV), wd.to(DEV)
    vec = [chan_vec(v) for v in (xsc, xsh, gamma, beta)]
    nb1 = l.pq_conv1x1_blocks_k(M, E, I, 0)
    slots = l.pq_bn_stat_slots(nb1)
    acc = torch.zeros(l.pq_bn_stat_bytes(ldp, slots) // 8, dtype=torch.int64, device=DEV)
    bad = torch.zeros(16, dtype=torch.int32, device=DEV)
    f = abi.Conv1x1Fwd(abi.Src(xd.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), I, 0), wed.data_ptr(), None, None, None,
                       M, I, E, 0, ldp, 1, 0, 0, 0, 0)
    f.stat_acc, f.stat_bad, f.stat_slots = acc.data_ptr(), bad.data_ptr(), slots
    abi.check(l.pq_conv1x1_fwd(C.byref(f), stream()))
    rm, rv = torch.zeros(E, device=DEV), torch.ones(E, device=DEV)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    outv = [torch.full((ldp + 16,), float('nan'), device=DEV) for _ in range(4)]
    out = torch.full((B, OH, OW, P), float('nan'), device=DEV)
    nblk = l.pq_xdw_fwd_blocks(B, H, W, I, E, stride)
    partial = torch.zeros(nblk * 2 * ldp, device=DEV)
    a = abi.XdwFwd()
    a.inp = f.inp
    a.we, a.wd = wed.data_ptr(), wdd.data_ptr()
    a.mid = abi.Src(None, None, None, 0, 2)
    a.mid.stat = abi.BnStat(acc.data_ptr(), bad.data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), rm.data_ptr(), rv.data_ptr(),
                            nbt.data_ptr(), *[v.data_ptr() for v in outv], slots, ldp, float(M), eps, mom, 1.0 / M)
    a.out, a.partial = out.data_ptr(), partial.data_ptr()
    a.B, a.IH, a.IW, a.OH, a.OW, a.I, a.E, a.stride = B, H, W, OH, OW, I, E, stride
    a.ldy, a.ldo, a.ldp = P, P, ldp
    abi.check(l.pq_xdw_fwd(C.byref(a), stream()))
    torch.cuda.synchronize()
    assert int(bad[0]) == 0
    close(outv[0][:E], mean.float(), what='mean')
    close(outv[1][:E], inv.float(), what='invstd')
    close(outv[2][:E], msc, what='scale')
    close(outv[3][:E], msh, what='shift')
    close(rm, (mom * mean).float(), what='running mean')
    close(rv, (0.9 + mom * var * M / (M - 1)).float(), what='running var')
    assert int(nbt[0]) == 1
    close(out, want, what='fused output with handed-over statistics')


@pytest.mark.parametrize('M,I,E,xact,affine', [(3000, 16, 96, 0, True), (5000, 24, 144, 0, False), (4097, 32, 192, 2, True),
                                                (70, 24, 40, 0, True), (262144, 24, 144, 0, True)])
def test_xstat_fwd_equals_batchnorm_statistics_of_the_expanded_tensor(M, I, E, xact, affine):
    """mean / biased variance of y = we x' from the moments of x' (sum x', sum x' x'^T) against nn.BatchNorm2d's statistics of y."""
    l = abi.lib()
    g = torch.Generator().manual_seed(M + I + E)
    x = rnd(M, I, g=g) + rnd(I, g=g)                              # channels with a mean: the cancellation case
    xsc, xsh = torch.rand(I, generator=g) + 0.5, rnd(I, g=g, scale=0.3)
    we = rnd(E, I, g=g, scale=1 / math.sqrt(I))
    gamma, beta = torch.rand(E, generator=g) + 0.5, rnd(E, g=g, scale=0.3)
    xin = x * xsc + xsh if affine else x
    if xact == 2:
        xin = xin.clamp(0, 6)
    y = xin.double() @ we.double().t()
    mean, var = y.mean(0), y.var(0, unbiased=False)
    eps, mom = 1e-5, 0.1
    inv = 1 / torch.sqrt(var + eps)
    xd, wed = x.to(DEV), we.to(DEV)
    vec = [chan_vec(v) for v in (xsc, xsh, gamma, beta)]
    rm, rv = torch.zeros(E, device=DEV), torch.ones(E, device=DEV)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    outv = [torch.full((E + 16,), float('nan'), device=DEV) for _ in range(4)]
    scratch = torch.zeros(l.pq_xstat_fwd_scratch_bytes(M, I) // 4, device=DEV)
    a = abi.XstatFwd()
    a.inp = abi.Src(xd.data_ptr(), vec[0].data_ptr() if affine else None, vec[1].data_ptr() if affine else None, I, xact)
    a.we, a.gamma, a.beta = wed.data_ptr(), vec[2].data_ptr(), vec[3].data_ptr()
    a.running_mean, a.running_var, a.num_batches_tracked = rm.data_ptr(), rv.data_ptr(), nbt.data_ptr()
    a.mean, a.invstd, a.scale, a.shift = [v.data_ptr() for v in outv]
    a.scratch, a.M, a.I, a.E, a.eps, a.momentum = scratch.data_ptr(), M, I, E, eps, mom
    abi.check(l.pq_xstat_fwd(C.byref(a), stream()))
    torch.cuda.synchronize()
    close(outv[0][:E], mean.float(), rt=1e-4, at=1e-5, what='mean')
    close(outv[1][:E], inv.float(), rt=1e-4, at=1e-5, what='invstd')
    close(outv[2][:E], (gamma.double() * inv).float(), rt=1e-4, at=1e-5, what='scale')
    close(outv[3][:E], (beta.double() - mean * gamma.double() * inv).float(), rt=1e-4, at=1e-4, what='shift')
    close(rm, (mom * mean).float(), rt=1e-4, at=1e-5, what='running mean')
    close(rv, (0.9 + mom * var * M / (M - 1)).float(), rt=1e-4, at=1e-5, what='running var')
    assert int(nbt[0]) == 1 and torch.isnan(outv[0][E:]).all()
    assert not bool(scratch.view(torch.int32).any()), 'the accumulators are not re-armed'


def test_xdw_fwd_refuses_what_it_does_not_cover():
    l = abi.lib()
    assert l.pq_xdw_fwd_blocks(2, 16, 16, 64, 384, 1) == 0            # contraction outside {16, 24, 32}
    assert l.pq_xdw_fwd_blocks(2, 15, 16, 16, 96, 2) == 0             # stride 2 on an odd plane
    x = torch.zeros(2 * 16 * 16 * 64, device=DEV)
    a = abi.XdwFwd()
    a.inp = abi.Src(x.data_ptr(), None, None, 64, 0)
    a.we = a.wd = a.out = x.data_ptr()
    a.mid = abi.Src(None, x.data_ptr(), x.data_ptr(), 0, 2)
    a.B, a.IH, a.IW, a.OH, a.OW, a.I, a.E, a.stride = 2, 16, 16, 16, 16, 64, 384, 1
    a.ldy, a.ldo, a.ldp = 384, 384, 384
    assert l.pq_xdw_fwd(C.byref(a), stream()) == -3 and b'pq_xdw_fwd' in l.pq_last_error()
    a.I, a.E, a.inp.ld, a.inp.act = 16, 96, 16, 2                     # activated input operand: not foldable
    assert l.pq_xdw_fwd(C.byref(a), stream()) == -3 and b'linear' in l.pq_last_error()
    f = abi.Conv1x1Fwd(abi.Src(x.data_ptr(), None, None, 64, 0), x.data_ptr(), None, None, None, 512, 64, 384, 0, 384, 1, 0, 0, 0, 0)
    assert l.pq_conv1x1_fwd(C.byref(f), stream()) == -1               # out == NULL without statistics
