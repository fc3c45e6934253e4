"""pq_reduce_batch: ONE launch sums the partial slabs of many weight-gradient entries (deferred form: dw == NULL).

(1) the kernel against fp64 sums on a table that exercises both lane counts, the bias column, a single split and the
    grouped-3x3 permutation; bit-identical on a second launch;
(2) each entry that can defer -- pq_conv1x1_wgrad (both kernel families), pq_conv1x1_bwd, pq_dw3x3_bwd (part 5 without
    part 6), pq_gconv3x3_wgrad -- gives, through pq_*_reduce_item + pq_reduce_batch, the gradient its own reduction gives
    (same partial slabs, another fixed order of additions: 1e-5 of the gradient's scale).
Work mirrored: the weight gradients torch.autograd accumulates for nn.Conv2d behind the reference's trainer.py:231-233."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from pqdet_amd import abi  # noqa: E402

DEV = 'cuda'


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run_batch(items):
    l = abi.lib()
    table = (abi.ReduceItem * len(items))(*items)
    nblk = l.pq_reduce_batch_plan(table, len(items))
    assert nblk > 0, l.pq_last_error()
    dev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(DEV)
    a = abi.ReduceBatch(dev.data_ptr(), len(items), nblk)
    abi.check(l.pq_reduce_batch(C.byref(a), stream()))
    torch.cuda.synchronize()
    return table


def same(got, want, what):
    scale = want.abs().max().item() + 1e-30
    err = (got.double() - want.double()).abs().max().item()
    assert err <= 1e-5 * scale, '%s: max err %g at scale %g' % (what, err, scale)


def test_table_of_mixed_items_matches_fp64_sums_and_replays_bit_identically():
    g = torch.Generator().manual_seed(5)
    items, checks = [], []
    # (splits, n0, n1, row pitch): 4 lanes; 16 lanes (> 64 splits); one split; ragged column counts
    for splits, n0, n1, pitch in ((5, 1000, 7, 1010), (200, 300, 0, 300), (1, 77, 0, 80), (64, 4097, 0, 4100), (65, 33, 5, 40),
                                  (12, 4096, 0, 4100)):
        slab = torch.randn(splits, pitch, generator=g).to(DEV)
        o0 = torch.full((n0,), float('nan'), device=DEV)
        o1 = torch.full((max(n1, 1),), float('nan'), device=DEV)
        it = abi.ReduceItem()
        it.slab, it.out0, it.out1 = slab.data_ptr(), o0.data_ptr(), (o1.data_ptr() if n1 else None)
        it.stride, it.splits, it.n0, it.n1 = pitch, splits, n0, n1
        items.append(it)
        want = slab.double().sum(0)
        checks.append((o0, want[:n0], 'plain n0=%d splits=%d' % (n0, splits), slab))
        if n1:
            checks.append((o1, want[n0:n0 + n1], 'bias column n1=%d' % n1, slab))
    # grouped 3x3: partial[g][split][o][tap * gw + i] -> dw[g * gw + o][i][tap]
    G, gw, ns = 3, 8, 9
    part = torch.randn(G, ns, gw, 9 * gw, generator=g).to(DEV)
    dw = torch.full((G * gw, gw, 9), float('nan'), device=DEV)
    it = abi.ReduceItem()
    it.slab, it.out0 = part.data_ptr(), dw.data_ptr()
    it.stride, it.gstride, it.splits, it.n0, it.ng, it.perm_gw = gw * 9 * gw, ns * gw * 9 * gw, ns, G * gw * 9 * gw, gw * 9 * gw, gw
    items.append(it)
    wantg = part.double().sum(1).reshape(G, gw, 9, gw).permute(0, 1, 3, 2).reshape(G * gw, gw, 9)
    checks.append((dw, wantg, 'grouped 3x3 permutation', part))
    table = run_batch(items)
    assert [t.lanes for t in table] == [4, 16, 4, 4, 16, 4, 4]
    assert [t.pad_ for t in table] == [1, 4, 1, 1, 1, 4, 1]          # float4 columns where pitches, counts and addresses allow
    first = []
    for got, want, what, _ in checks:
        same(got[:want.shape[0]] if got.dim() == 1 else got, want, what)
        first.append(got.clone())
    run_batch(items)
    for (got, _, what, _), f in zip(checks, first):
        assert torch.equal(got, f), what + ': second launch differs'


def test_plan_refuses_an_incomplete_item():
    it = abi.ReduceItem()
    table = (abi.ReduceItem * 1)(it)
    assert abi.lib().pq_reduce_batch_plan(table, 1) == -1
    assert b'incomplete' in abi.lib().pq_last_error()
    a = abi.ReduceBatch(None, 0, 0)
    assert abi.lib().pq_reduce_batch(C.byref(a), stream()) != 0


def vec(n, g, lo=0.5):
    return (torch.rand(n + 32, generator=g) + lo).to(DEV)


@pytest.mark.parametrize('M,I,O,bias', [(4096, 160, 960, False), (16384, 352, 256, False), (700, 1024, 75, True), (5000, 96, 24, False)])
def test_conv1x1_wgrad_deferred_equals_its_own_reduction(M, I, O, bias):
    g = torch.Generator().manual_seed(M + O)
    l = abi.lib()
    ldg = (O + 3) // 4 * 4
    gy, y, x = [torch.randn(M, c, generator=g).to(DEV) for c in (ldg, ldg, I)]
    v = [vec(O, g) for _ in range(4)]
    xs = [vec(I, g), vec(I, g, -0.5)]
    scratch = torch.empty(l.pq_conv1x1_wgrad_scratch_bytes(M, I, O) // 4 + 16, device=DEV)
    dw1, dw2 = [torch.full((O, I), float('nan'), device=DEV) for _ in range(2)]
    db1, db2 = [torch.full((O,), float('nan'), device=DEV) for _ in range(2)]
    a = abi.Conv1x1Wgrad()
    a.dy = abi.GSrc(gy.data_ptr(), y.data_ptr(), *[t.data_ptr() for t in v], ldg, ldg)
    a.inp = abi.Src(x.data_ptr(), xs[0].data_ptr(), xs[1].data_ptr(), I, 2)
    a.dw, a.dbias, a.scratch = dw1.data_ptr(), (db1.data_ptr() if bias else None), scratch.data_ptr()
    a.M, a.I, a.O = M, I, O
    abi.check(l.pq_conv1x1_wgrad(C.byref(a), stream()))
    a.dw, a.dbias = dw2.data_ptr(), (db2.data_ptr() if bias else None)
    it = abi.ReduceItem()
    abi.check(l.pq_conv1x1_wgrad_reduce_item(C.byref(a), C.byref(it)))
    scratch.fill_(float('nan'))
    a.dw = None
    abi.check(l.pq_conv1x1_wgrad(C.byref(a), stream()))
    torch.cuda.synchronize()
    assert bool(torch.isnan(dw2).all())                   # the deferred call wrote no gradient
    run_batch([it])
    same(dw2, dw1, 'weight gradient')
    if bias:
        same(db2, db1, 'bias gradient')


@pytest.mark.parametrize('M,I,O', [(40000, 16, 96), (30000, 144, 24), (20000, 32, 16)])
def test_conv1x1_bwd_deferred_equals_its_own_reduction(M, I, O):
    g = torch.Generator().manual_seed(M + I)
    l = abi.lib()
    nb = l.pq_conv1x1_bwd_blocks(M, I, O)
    assert nb > 0
    gy, y, x = [torch.randn(M, c, generator=g).to(DEV) for c in (O, O, I)]
    w = (torch.randn(O, I, generator=g) * 0.2).to(DEV)
    vO, vI = [vec(O, g) for _ in range(4)], [vec(I, g) for _ in range(4)]
    gx = torch.empty(M, I, device=DEV)
    part = torch.empty(nb * 2 * 32 * ((I + 31) // 32), device=DEV)
    scratch = torch.empty(l.pq_conv1x1_bwd_scratch_bytes(M, I, O) // 4 + 16, device=DEV)
    dw1, dw2 = [torch.full((O, I), float('nan'), device=DEV) for _ in range(2)]
    ldp = (I + 31) // 32 * 32
    a = abi.Conv1x1Bwd(abi.GSrc(gy.data_ptr(), y.data_ptr(), *[t.data_ptr() for t in vO], O, O), w.data_ptr(),
                       abi.Src(x.data_ptr(), vI[0].data_ptr(), vI[1].data_ptr(), I, 0),
                       abi.GSink(gx.data_ptr(), x.data_ptr(), vI[0].data_ptr(), vI[1].data_ptr(), vI[2].data_ptr(), vI[3].data_ptr(), part.data_ptr(),
                                 I, I, 0, 0, 1, ldp),
                       dw1.data_ptr(), scratch.data_ptr(), M, I, O)
    abi.check(l.pq_conv1x1_bwd(C.byref(a), stream()))
    torch.cuda.synchronize()
    gx1 = gx.clone()
    a.dw = dw2.data_ptr()
    it = abi.ReduceItem()
    abi.check(l.pq_conv1x1_bwd_reduce_item(C.byref(a), C.byref(it)))
    a.dw = None
    scratch.fill_(float('nan'))
    abi.check(l.pq_conv1x1_bwd(C.byref(a), stream()))
    torch.cuda.synchronize()
    assert torch.equal(gx, gx1) and bool(torch.isnan(dw2).all())
    run_batch([it])
    same(dw2, dw1, 'weight gradient of the fused narrow backward')


@pytest.mark.parametrize('B,H,W,Cc,s', [(4, 64, 64, 96, 1), (2, 256, 256, 32, 1), (4, 64, 64, 144, 2)])
def test_dw3x3_bwd_without_part6_equals_part6(B, H, W, Cc, s):
    g = torch.Generator().manual_seed(H + Cc)
    l = abi.lib()
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    nb = l.pq_dw3x3_bwd_walk_blocks(B, H, W, Cc, s)
    yin = torch.randn(B, H, W, Cc, generator=g).to(DEV)
    gin = torch.empty(B, H, W, Cc, device=DEV)
    yd, gd = [torch.randn(B, OH, OW, Cc, generator=g).to(DEV) for _ in range(2)]
    wd = torch.randn(Cc, 9, generator=g).to(DEV)
    v = [vec(Cc, g) for _ in range(8)]
    part = torch.empty(nb * 2 * Cc, device=DEV)
    scratch = torch.empty(nb * Cc * 9 + 16, device=DEV)
    dw1, dw2 = [torch.full((Cc, 9), float('nan'), device=DEV) for _ in range(2)]
    f5 = abi.DwBwd(abi.GSrc(gd.data_ptr(), yd.data_ptr(), *[t.data_ptr() for t in v[:4]], Cc, Cc),
                   abi.Src(yin.data_ptr(), v[4].data_ptr(), v[5].data_ptr(), Cc, 2), wd.data_ptr(),
                   abi.GSink(gin.data_ptr(), yin.data_ptr(), v[4].data_ptr(), v[5].data_ptr(), v[6].data_ptr(), v[7].data_ptr(), part.data_ptr(),
                             Cc, Cc, 2, 0, 1, Cc),
                   None, scratch.data_ptr(), B, H, W, OH, OW, Cc, s, 5)
    abi.check(l.pq_dw3x3_bwd(C.byref(f5), stream()))
    torch.cuda.synchronize()
    rows = scratch.clone()                              # part 6 of a long list sums slices of it in place
    f6 = abi.DwBwd(abi.GSrc(), abi.Src(), None, abi.GSink(), dw2.data_ptr(), scratch.data_ptr(), B, H, W, OH, OW, Cc, s, 6)
    it = abi.ReduceItem()
    abi.check(l.pq_dw3x3_bwd_reduce_item(C.byref(f6), C.byref(it)))
    assert it.splits == nb and it.n0 == Cc * 9
    run_batch([it])
    f6.dw = dw1.data_ptr()
    abi.check(l.pq_dw3x3_bwd(C.byref(f6), stream()))
    torch.cuda.synchronize()
    same(dw2, dw1, 'depthwise weight gradient')
    same(dw2, rows[:nb * Cc * 9].reshape(nb, Cc, 9).double().sum(0).float(), 'depthwise weight gradient against fp64 sums')


@pytest.mark.parametrize('B,H,W,Cc,gw,s', [(4, 32, 32, 48, 24, 1), (4, 32, 32, 64, 16, 2), (8, 64, 64, 48, 24, 2)])
def test_gconv3x3_wgrad_deferred_equals_its_own_reduction(B, H, W, Cc, gw, s):
    g = torch.Generator().manual_seed(H + Cc + s)
    l = abi.lib()
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(B, H, W, Cc, generator=g).to(DEV)
    yd, gd = [torch.randn(B, OH, OW, Cc, generator=g).to(DEV) for _ in range(2)]
    v = [vec(Cc, g) for _ in range(6)]
    scratch = torch.empty(l.pq_gconv3x3_wgrad_scratch_bytes(B, OH, OW, Cc, gw) // 4 + 16, device=DEV)
    dw1, dw2 = [torch.full((Cc, gw, 3, 3), float('nan'), device=DEV) for _ in range(2)]
    a = abi.GconvWgrad(abi.GSrc(gd.data_ptr(), yd.data_ptr(), *[t.data_ptr() for t in v[:4]], Cc, Cc),
                       abi.Src(x.data_ptr(), v[4].data_ptr(), v[5].data_ptr(), Cc, 1), dw1.data_ptr(), scratch.data_ptr(),
                       B, H, W, OH, OW, Cc, gw, s)
    abi.check(l.pq_gconv3x3_wgrad(C.byref(a), stream()))
    a.dw = dw2.data_ptr()
    it = abi.ReduceItem()
    abi.check(l.pq_gconv3x3_wgrad_reduce_item(C.byref(a), C.byref(it)))
    a.dw = None
    scratch.fill_(float('nan'))
    abi.check(l.pq_gconv3x3_wgrad(C.byref(a), stream()))
    torch.cuda.synchronize()
    assert bool(torch.isnan(dw2).all())
    run_batch([it])
    same(dw2, dw1, 'grouped 3x3 weight gradient')

