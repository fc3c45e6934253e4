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
    (2, 24, 24, 16, 96, 1, 0),
    (1, 20, 36, 24, 144, 2, 0),
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


@pytest.mark.parametrize('B,H,W,I,E,stride', [(2, 32, 32, 24, 144, 1), (2, 32, 32, 16, 96, 2), (2, 16, 48, 24, 96, 1)])
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


@pytest.mark.parametrize('M,I,E,xact,affine', [(3000, 16, 96, 0, True), (5000, 24, 144, 0, False), (4097, 24, 192, 2, True),
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


@pytest.mark.parametrize('B,H,W,I,E,stride,affine,accumulate', [
    (2, 16, 32, 24, 144, 1, True, 0),
    (1, 22, 24, 24, 144, 1, False, 1),      # rows that are not a whole number of steps; residual block: running gradient in the sink
    (1, 22, 24, 24, 144, 1, True, 2),       # ... the running gradient lives in ANOTHER tensor (pq_gsink.addend: folded shortcut)
    (2, 16, 32, 16, 96, 2, True, 0),
    (1, 24, 40, 24, 144, 2, True, 0),
    (2, 34, 20, 24, 144, 1, True, 0),       # partial strip (20 columns), several segments
    (1, 128, 128, 24, 144, 1, True, 0),     # benchmark-step plane (one image)
])
def test_xdw_bwd_matches_autograd_through_expand_batchnorm_relu6_depthwise(B, H, W, I, E, stride, affine, accumulate):
    """pq_xdw_bwd (recompute + analytic BatchNorm-backward fold) against torch autograd (fp64) through
    x' -> conv1x1 -> BatchNorm(train) -> ReLU6 -> depthwise 3x3; the sink of the block input with its own statistics."""
    l = abi.lib()
    g, x, xsc, xsh, we, wd = _case(B, H, W, I, E, stride, seed=11)
    if not affine:
        xsc, xsh = torch.ones(I), torch.zeros(I)
    gamma, beta = torch.rand(E, generator=g) + 0.5, rnd(E, g=g, scale=0.5) + 0.5
    eps = 1e-5
    M = B * H * W
    xp = (x.double() * xsc.double() + xsh.double()).requires_grad_(True)
    wed, wdd = we.double().requires_grad_(True), wd.double().requires_grad_(True)
    gm, bt = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ye = xp @ wed.t()
    mean, var = ye.view(M, E).mean(0), ye.view(M, E).var(0, unbiased=False)
    inv = 1 / torch.sqrt(var + eps)
    u = (ye - mean) * inv * gm + bt
    u.retain_grad()
    h = u.clamp(0, 6)
    yd = F.conv2d(h.permute(0, 3, 1, 2), wdd, stride=stride, padding=1, groups=E).permute(0, 2, 3, 1)
    OH, OW = yd.shape[1:3]
    gd = rnd(B, OH, OW, E, g=g)
    pd, qd = torch.rand(E, generator=g) + 0.5, rnd(E, g=g, scale=0.05)
    mud, rd = yd.detach().view(-1, E).mean(0).float(), rnd(E, g=g, scale=0.05)
    dyp = pd.double() * gd.double() + qd.double() * (yd.detach() - mud.double()) + rd.double()
    (yd * dyp).sum().backward()
    prev = rnd(B, H, W, I, g=g) if accumulate else None
    gx_want = xp.grad + (prev.double() if accumulate else 0)                      # linear operand: act' = 1
    mean_x = x.view(M, I).double().mean(0)
    inv_x = 1 / torch.sqrt(x.view(M, I).double().var(0, unbiased=False) + eps)
    xd, wedev, wddev = x.to(DEV), we.to(DEV), wd.to(DEV)
    ydd, gdd = yd.detach().float().to(DEV).contiguous(), gd.to(DEV)
    vec = lambda v: chan_vec(v.detach().float())
    v_x = [vec(xsc), vec(xsh), vec(mean_x), vec(inv_x)]
    v_e = [vec(gm * inv), vec(bt - mean * gm * inv), vec(mean), vec(inv), vec(gamma)]
    v_d = [vec(pd), vec(qd), vec(mud), vec(rd)]
    mom = torch.cat([(xp.detach().view(M, I).t() @ xp.detach().view(M, I)).reshape(-1), xp.detach().view(M, I).sum(0)]).to(DEV)
    assert mom.dtype == torch.float64
    nfix = l.pq_xdw_bwd_fix_blocks(M, I)
    ldp = 32
    gdst = prev.to(DEV).clone() if accumulate == 1 else torch.full((B, H, W, I), float('nan'), device=DEV)
    addend = prev.to(DEV).clone() if accumulate == 2 else None
    partial = torch.full((nfix * 2 * ldp,), float('nan'), device=DEV)
    assert l.pq_xdw_bwd_blocks(B, H, W, I, E, stride) > 0
    scratch = torch.full((l.pq_xdw_bwd_scratch_bytes(B, H, W, I, E, stride) // 4,), float('nan'), device=DEV)
    outs = [torch.full((E * I,), float('nan'), device=DEV), torch.full((E * 9,), float('nan'), device=DEV),
            torch.full((E,), float('nan'), device=DEV), torch.full((E,), float('nan'), device=DEV)]
    gexp = torch.full((B, H, W, E), float('nan'), device=DEV)
    a = abi.XdwBwd()
    a.inp = abi.Src(xd.data_ptr(), v_x[0].data_ptr() if affine else None, v_x[1].data_ptr() if affine else None, I, 0)
    a.we, a.wd = wedev.data_ptr(), wddev.data_ptr()
    a.mid_scale, a.mid_shift, a.mid_mean, a.mid_invstd, a.mid_gamma = [v.data_ptr() for v in v_e]
    a.mid_act = 2
    a.dy = abi.GSrc(gdd.data_ptr(), ydd.data_ptr(), *[v.data_ptr() for v in v_d], E, E)
    a.dx = abi.GSink(gdst.data_ptr(), xd.data_ptr(), v_x[0].data_ptr() if affine else None, v_x[1].data_ptr() if affine else None,
                     v_x[2].data_ptr(), v_x[3].data_ptr(), partial.data_ptr(), I, I, 0, int(accumulate > 0), 1, ldp)
    a.dx.addend = abi.ptr(addend)
    a.dwe, a.dwd, a.dgamma, a.dbeta = [o.data_ptr() for o in outs]
    a.moments, a.scratch, a.gexp, a.ldge = mom.data_ptr(), scratch.data_ptr(), gexp.data_ptr(), E
    a.B, a.IH, a.IW, a.OH, a.OW, a.I, a.E, a.stride = B, H, W, OH, OW, I, E, stride
    abi.check(l.pq_xdw_bwd(C.byref(a), stream()))
    torch.cuda.synchronize()
    close(gexp, u.grad.float(), what='gradient w.r.t. the BatchNorm output of the expanded tensor')
    close(outs[1].view(E, 1, 3, 3), wdd.grad.float(), rt=1e-3, at=1e-3, what='depthwise weight gradient')
    close(outs[3], bt.grad.float(), rt=1e-3, at=1e-3, what='dbeta')
    close(outs[2], gm.grad.float(), rt=1e-3, at=1e-3, what='dgamma')
    close(outs[0].view(E, I), wed.grad.float(), rt=1e-3, at=1e-3, what='expand weight gradient')
    close(gdst, gx_want.float(), what='block input gradient')
    s1, s2 = reduce_partial(partial, nfix, ldp, I)
    xh = (x.double() - mean_x) * inv_x
    # (the first sum is ~0 by construction -- BatchNorm's backward removes the mean of the gradient -- so both are judged against
    # the size of their summands, not of the result)
    for got, terms, what in ((s1, gx_want, 'sink sum'), (s2, gx_want * xh, 'sink sum xhat')):
        err = (got - terms.sum((0, 1, 2))).abs()
        assert bool((err <= 1e-5 * terms.abs().sum((0, 1, 2)) + 1e-6).all()), '%s: max err %g' % (what, err.max().item())


def test_xstat_fwd_takes_the_block_inputs_batchnorm_from_its_producers_accumulators():
    """pq_conv1x1_fwd(stat_acc) -> pq_xstat_fwd(in.stat): the block input's own train-mode BatchNorm is handed over (no finalize launch);
    the moments pass derives scale / shift per lane, its first workgroup stores the vectors and the running statistics."""
    l = abi.lib()
    g = torch.Generator().manual_seed(5)
    M, C0, I, E = 6000, 32, 16, 96
    x0, w0 = rnd(M, C0, g=g), rnd(I, C0, g=g, scale=0.2)
    gam, bet = torch.rand(I, generator=g) + 0.5, rnd(I, g=g, scale=0.3)
    we = rnd(E, I, g=g, scale=0.25)
    ge, be = torch.rand(E, generator=g) + 0.5, rnd(E, g=g, scale=0.3)
    y0 = x0.double() @ w0.double().t()
    m0, v0 = y0.mean(0), y0.var(0, unbiased=False)
    xin = (y0 - m0) / torch.sqrt(v0 + 1e-5) * gam.double() + bet.double()          # the block input x'
    ye = xin @ we.double().t()
    me, ve = ye.mean(0), ye.var(0, unbiased=False)
    dev = [t.to(DEV) for t in (x0, w0, we)]
    vec = [chan_vec(v) for v in (gam, bet, ge, be)]
    ldp = 16
    nb = l.pq_conv1x1_blocks_k(M, I, C0, 0)
    slots = l.pq_bn_stat_slots(nb)
    acc = torch.zeros(l.pq_bn_stat_bytes(ldp, slots) // 8, dtype=torch.int64, device=DEV)
    bad = torch.zeros(16, dtype=torch.int32, device=DEV)
    raw = torch.full((M, I), float('nan'), device=DEV)
    f = abi.Conv1x1Fwd(abi.Src(dev[0].data_ptr(), None, None, C0, 0), dev[1].data_ptr(), None, raw.data_ptr(), None, M, C0, I, I, ldp, 1, 0, 0, 0, 0)
    f.stat_acc, f.stat_bad, f.stat_slots = acc.data_ptr(), bad.data_ptr(), slots
    abi.check(l.pq_conv1x1_fwd(C.byref(f), stream()))
    rm, rv = torch.zeros(I, device=DEV), torch.ones(I, device=DEV)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    inv = [torch.full((32,), float('nan'), device=DEV) for _ in range(4)]          # mean, invstd, scale, shift of the block input
    outv = [torch.full((E + 16,), float('nan'), device=DEV) for _ in range(4)]
    scratch = torch.zeros(l.pq_xstat_fwd_scratch_bytes(M, I) // 4, device=DEV)
    a = abi.XstatFwd()
    a.inp = abi.Src(raw.data_ptr(), inv[2].data_ptr(), inv[3].data_ptr(), I, 0)
    a.inp.stat = abi.BnStat(acc.data_ptr(), bad.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(),
                            *[v.data_ptr() for v in inv], slots, ldp, float(M), 1e-5, 0.1, 1.0 / M)
    a.we, a.gamma, a.beta = dev[2].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr()
    a.mean, a.invstd, a.scale, a.shift = [v.data_ptr() for v in outv]
    a.scratch, a.M, a.I, a.E, a.eps, a.momentum = scratch.data_ptr(), M, I, E, 1e-5, 0.1
    assert l.pq_bn_stat_ok(abi.OP_XSTAT_FWD, C.byref(a), 1) == 1
    abi.check(l.pq_xstat_fwd(C.byref(a), stream()))
    torch.cuda.synchronize()
    close(inv[0][:I], m0.float(), rt=1e-4, at=1e-5, what='block input mean')
    close(inv[2][:I], (gam.double() / torch.sqrt(v0 + 1e-5)).float(), rt=1e-4, at=1e-5, what='block input scale')
    close(rm, (0.1 * m0).float(), rt=1e-4, at=1e-5, what='running mean of the block input')
    assert int(nbt[0]) == 1
    close(outv[0][:E], me.float(), rt=1e-4, at=1e-4, what='expanded mean')
    close(outv[1][:E], (1 / torch.sqrt(ve + 1e-5)).float(), rt=1e-4, at=1e-5, what='expanded invstd')


def test_xdw_fwd_refuses_what_it_does_not_cover():
    l = abi.lib()
    assert l.pq_xdw_fwd_blocks(2, 16, 16, 64, 384, 1) == 0            # contraction outside {16, 24}
    assert l.pq_xdw_fwd_blocks(2, 16, 16, 32, 192, 1) == 0 and l.pq_xdw_bwd_blocks(2, 16, 16, 32, 192, 1) == 0
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
