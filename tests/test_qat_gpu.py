"""Quantisation-aware training (BASELINE config 5) on the HIP engine.

Fake-quantisation makes the network a step function of its inputs: one value that lands on the other side of a rounding
boundary moves by a whole quantisation step (0.4 % of the tensor's range), and the box-regression loss of a responsible
cell sees exp() of it.  The same oracle code run on two different CPUs (this container, the GPU box's host) already
differs by 0.1-1.8 % in the total loss, 4-12 % in the box term and up to 33 % in single observer ranges at 256x256, and by
12 % in the total loss at 64x64 (scripts/qat_noise_probe.py).  So parity is established in
three layers:
  1. the kernels (weight preparation / gradient, activation observer + quantiser forward / backward) against the oracle's
     functions on random tensors -- exact;
  2. the whole QAT step with the rounding switched off (observers on, fake_quant off): a continuous function that
     exercises every piece of QAT plumbing (weight scaling by gamma/sigma, BatchNorm on conv/sf, observer updates, the
     extra gradient paths) -- compared tightly with the oracle in the same mode;
  3. the full QAT step against the G9 fixtures recorded from the reference -- observer ranges and losses within the
     cross-machine noise band measured above.
"""
import ctypes as C
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import qat as oqat  # noqa: E402
from pqdet_amd import abi, optim, tools  # noqa: E402
from pqdet_amd.model import zoo  # noqa: E402
from pqdet_amd.model.interpreter import DetectionModel  # noqa: E402
from seedfill import seeded_fill, seeded_images  # noqa: E402
from test_oracle_qat import _g9, _oracle_at_g9b, qat_batch  # noqa: E402

DEV = 'cuda'


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _product(seed=0):
    net = DetectionModel(io.StringIO(zoo.cfg_text('mobilenetv2-fpn')), quant=True)
    seeded_fill(net, seed)
    net = net.to(DEV)
    net.train()
    tools.fuse_model(net)
    tools.prepare_qat(net, backend='qnnpack')
    return net


# ------------------------------------------------------------------------------------------------- 1. kernels
@pytest.mark.parametrize('rows,Cc,act', [(300, 32, 1), (1000, 75, 0), (64, 96, 1)])
def test_qact_forward_backward_match_oracle_exactly(rows, Cc, act):
    g = torch.Generator().manual_seed(rows + Cc)
    l = abi.lib()
    ld = (Cc + 3) // 4 * 4
    x = torch.zeros(rows, ld)
    x[:, :Cc] = torch.randn(rows, Cc, generator=g) * 3
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.5
    fq = oqat.act_fq()
    state = torch.tensor([float('inf'), float('-inf'), 1.0, 0.0], device=DEV)
    xd = x.to(DEV)
    scd = torch.zeros(ld + 16, device=DEV); scd[:Cc] = sc.to(DEV)
    shd = torch.zeros(ld + 16, device=DEV); shd[:Cc] = sh.to(DEV)
    out = torch.zeros(rows, ld, device=DEV)
    part = torch.empty(2 * l.pq_qact_blocks(rows * Cc) + 8, device=DEV)
    for step in range(3):                                   # moving average over several "batches"
        xs = (x * (1 + 0.3 * step))
        xd.copy_(xs)
        u = (xs[:, :Cc] * sc + sh).requires_grad_(True)
        v = torch.relu(u) if act else u
        want = fq(v)
        a = abi.Qact(abi.Src(xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), ld, act), out.data_ptr(), state.data_ptr(),
                     part.data_ptr(), rows, Cc, ld, 0, 1, 1, 0.01)
        abi.check(l.pq_qact_fwd(C.byref(a), _stream()))
        torch.cuda.synchronize()
        st = state.cpu()
        # u = scale*x + shift is one FMA on the device and two roundings on the host: ranges agree to an ulp
        np.testing.assert_allclose(st[:3].numpy(), [fq.min_val.item(), fq.max_val.item(), fq.scale.item()], rtol=2e-6, atol=1e-7)
        assert st[3].item() == float(fq.zero_point.item())
        # ... and a handful of values may land on the other side of a rounding boundary (one quantisation step)
        diff = (out[:, :Cc].cpu() - want.detach()).abs()
        assert (diff > 1e-6).float().mean().item() < 2e-3 and diff.max().item() <= fq.scale.item() * 1.001
    # backward of the last step: straight-through mask x relu'
    gq = torch.randn(rows, Cc, generator=g)
    want.backward(gq)
    gwant = u.grad * (sc if False else 1.0)                 # gradient w.r.t. u (the pre-activation)
    gqd = torch.zeros(rows, ld, device=DEV); gqd[:, :Cc] = gq.to(DEV)
    gout = torch.full((rows, ld), float('nan'), device=DEV)
    Cq = Cc if Cc % 4 == 0 else ld
    b = abi.QactBwd(gqd.data_ptr(), abi.GSink(gout.data_ptr(), xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), None, None, None,
                                               ld, ld, act, 0, 1, 0), state.data_ptr(), rows, Cq, ld, 1)
    abi.check(l.pq_qact_bwd(C.byref(b), _stream()))
    torch.cuda.synchronize()
    d = (gout[:, :Cc].cpu() - gwant).abs()
    assert (d > 1e-6).float().mean().item() < 2e-3          # identical except where u sits on a rounding / ReLU boundary


def test_gradient_sinks_apply_the_straight_through_mask():
    """Fused backward: a consumer's sink with qstate set finalises g = dz * act'(u) * [fq(act(u)) inside the range], both in the
    element-wise sinks (pq_route_bwd) and in the GEMM's row-major sink epilogue (pq_conv1x1_dgrad)."""
    g = torch.Generator().manual_seed(21)
    l = abi.lib()
    rows, Cc = 512, 64
    y = torch.randn(rows, Cc, generator=g) * 2
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    # observer state with a range that clips the top of the distribution
    state = torch.tensor([0.0, 2.0, 2.0 / 255, 0.0])
    u = (y * sc + sh).requires_grad_(True)
    v = torch.relu(u)
    q = oqat._FakeQuantSTE.apply(v, float(state[2]), int(state[3]), 0, 255)
    up = torch.randn(rows, Cc, generator=g)
    q.backward(up)
    want = u.grad
    assert (want == 0).float().mean() > 0.3 and ((v > 2.01) & (up != 0)).any()       # both ReLU and range clipping occur
    yd, upd, st = y.to(DEV), up.to(DEV), state.to(DEV)
    scd, shd = sc.to(DEV), sh.to(DEV)
    gout = torch.full((rows, Cc), float('nan'), device=DEV)
    sink = abi.GSink(gout.data_ptr(), yd.data_ptr(), scd.data_ptr(), shd.data_ptr(), None, None, None, Cc, Cc, 1, 0, 1, 0, st.data_ptr())
    r = abi.RouteBwd(upd.data_ptr(), sink, 1, rows, 1, Cc, Cc, 0, 1, None, 0, 0, 1.0)
    abi.check(l.pq_route_bwd(C.byref(r), _stream()))
    torch.cuda.synchronize()
    d = (gout.cpu() - want).abs()
    assert (d > 1e-6).float().mean().item() < 2e-3
    # GEMM sink: dz = dy @ w (identity-like weights so that dz == dy), same finalisation
    O_ = Cc
    w = torch.eye(O_, Cc)
    gout2 = torch.full((rows, Cc), float('nan'), device=DEV)
    a = abi.Conv1x1Dgrad()
    a.dy = abi.GSrc(upd.data_ptr(), None, None, None, None, None, O_, O_)
    wd = w.to(DEV)
    a.w = wd.data_ptr()
    a.dx = abi.GSink(gout2.data_ptr(), yd.data_ptr(), scd.data_ptr(), shd.data_ptr(), None, None, None, Cc, Cc, 1, 0, 1, 0, st.data_ptr())
    a.M, a.I, a.O = rows, Cc, O_
    abi.check(l.pq_conv1x1_dgrad(C.byref(a), _stream()))
    torch.cuda.synchronize()
    d2 = (gout2.cpu() - want).abs()
    assert (d2 > 1e-5).float().mean().item() < 2e-3


def test_weight_prep_and_weight_grad_match_oracle():
    g = torch.Generator().manual_seed(9)
    l = abi.lib()
    shapes = [(32, 27, True), (96, 16, True), (75, 1024, False), (144, 9, True)]
    items, keep, refs = [], [], []
    for O_, K_, has_bn in shapes:
        w = torch.randn(O_, K_, generator=g) * 0.2
        gamma = torch.rand(O_, generator=g) + 0.5
        rv = torch.rand(O_, generator=g) + 0.3
        dwq = torch.randn(O_, K_, generator=g)
        t = dict(w=w.to(DEV), gamma=gamma.to(DEV), rv=rv.to(DEV), wq=torch.empty(O_, K_, device=DEV), sf=torch.empty(O_ + 16, device=DEV),
                 state=torch.tensor([float('inf'), float('-inf'), 1.0, 0.0], device=DEV), dwq=dwq.to(DEV),
                 dw=torch.full((O_, K_), float('nan'), device=DEV), dgamma=torch.ones(O_, device=DEV))
        keep.append(t)
        items.append(abi.QatWeightItem(t['w'].data_ptr(), t['gamma'].data_ptr() if has_bn else None, t['rv'].data_ptr() if has_bn else None,
                                       t['wq'].data_ptr(), t['sf'].data_ptr(), t['state'].data_ptr(), t['dwq'].data_ptr(),
                                       t['dw'].data_ptr(), t['dgamma'].data_ptr() if has_bn else None, 1e-5, O_, K_))
        refs.append((w, gamma, rv, dwq, has_bn))
    table = (abi.QatWeightItem * len(items))(*items)
    dev_table = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(DEV)
    args = abi.QatWeight(dev_table.data_ptr(), len(items), 1, 1, 0.01)
    abi.check(l.pq_qat_weight_prep(C.byref(args), _stream()))
    abi.check(l.pq_qat_weight_grad(C.byref(args), _stream()))
    torch.cuda.synchronize()
    for t, (w, gamma, rv, dwq, has_bn) in zip(keep, refs):
        wr = w.clone().requires_grad_(True)
        gr = gamma.clone().requires_grad_(True)
        fq = oqat.weight_fq()
        sf = gr / torch.sqrt(rv + 1e-5) if has_bn else torch.ones_like(gr)
        wq = fq(wr * sf.reshape(-1, 1))
        # the conv output is divided by sf again: d/dsf of that division is -sum(dwq * wq)/sf per channel
        loss = (wq * dwq).sum() - ((wq.detach() * dwq).sum(1) / sf.detach() * sf).sum() if has_bn else (wq * dwq).sum()
        loss.backward()
        st = t['state'].cpu()
        assert st[2].item() == fq.scale.item() and st[0].item() == fq.min_val.item()
        assert torch.equal(t['wq'].cpu(), wq.detach())
        assert torch.allclose(t['dw'].cpu(), wr.grad, rtol=1e-6, atol=1e-7)
        if has_bn:
            assert torch.allclose(t['dgamma'].cpu() - 1.0, gr.grad, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------------- 2. rounding off
def test_qat_plumbing_with_rounding_switched_off_matches_oracle():
    text = zoo.cfg_text('mobilenetv2-fpn')
    net = _product(seed=4)
    net.qat_fake_quant_enabled = False
    ref = oqat.OracleQATNet(io.StringIO(text))
    seeded_fill(ref.float_net, 4)
    ref.train()
    for m in ref.modules():
        if isinstance(m, oqat.FakeQuant):
            m.fake_quant_enabled = False
    for step in range(2):
        x, tgt = qat_batch(step, B=3, S=160)
        net.zero_grad()
        out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
        out['loss'].mean().backward()
        ref.zero_grad()
        want = ref(x, tgt)
        want['loss'].mean().backward()
        for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss'):
            assert abs(out[k].item() - want[k].item()) <= 1e-3 * abs(want[k].item()) + 1e-5, (step, k)
    mine = dict(net.named_parameters())
    bad = []
    top = max(p.grad.double().norm().item() for p in ref.float_net.parameters())
    for k, p in ref.float_net.named_parameters():
        gw = p.grad.double()
        if gw.norm().item() < 1e-6 * top:      # e.g. the BatchNorm shift of a linear layer that feeds only conv+BN: exactly 0
            continue                           # in exact arithmetic, pure round-off in any fp32 evaluation
        e = ((mine[k].grad.cpu().double() - gw).norm() / gw.norm()).item()
        if e > 0.08:                       # fp32 conditioning of deep train-mode BN stacks, see test_model_gpu.py
            bad.append((k, e))
    assert len(bad) <= 3, bad
    # observers saw the same ranges, running statistics moved the same way
    from test_oracle_qat import qat_observers
    want_obs = qat_observers(ref)
    got_obs = net._qat_state.cpu().numpy()
    np.testing.assert_allclose(got_obs[:, :3], want_obs[:, :3], rtol=2e-3, atol=1e-5)
    sd, rd = net.plain_state_dict(), ref.float_net.state_dict()
    for k in rd:
        if 'running' in k:
            np.testing.assert_allclose(sd[k].cpu().numpy(), rd[k].numpy(), rtol=2e-3, atol=1e-5, err_msg=k)


# ------------------------------------------------------------------------------------------------- 3. full QAT vs G9
def test_qat_two_steps_against_reference_g9_within_quantisation_noise():
    g, names = _g9()
    net = _product()
    losses = []
    for step in range(2):
        x, tgt = qat_batch(step)
        net.zero_grad()
        out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
        out['loss'].mean().backward()
        losses.append([out[k].item() for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')]
                      + [v.item() for v in out['loss_per_branch']])
    net.check_nan()
    losses = np.array(losses)
    print('losses', losses, 'reference', g['g9_loss'])
    np.testing.assert_allclose(losses[:, 0], g['g9_loss'][:, 0], rtol=0.03)           # total
    np.testing.assert_allclose(losses[:, 2], g['g9_loss'][:, 2], rtol=0.03)           # objectness (dominant, smooth)
    np.testing.assert_allclose(losses[:, 3], g['g9_loss'][:, 3], rtol=0.10)           # class
    assert np.all(np.abs(losses[:, 1] - g['g9_loss'][:, 1]) <= 0.5 * g['g9_loss'][:, 1])   # box term: exp() of 1-LSB flips
    obs = net.qat_observers()
    got = np.array([obs[k] for k in names['observers']])
    want = g['g9_observers']
    rng_w = np.maximum(np.abs(want[:, 0]), np.abs(want[:, 1]))
    dev = np.abs(got[:, :2] - want[:, :2]).max(1) / (rng_w + 1e-6)
    off = [(names['observers'][i], got[i, :2].tolist(), want[i, :2].tolist()) for i in np.argsort(-dev)[:6] if dev[i] > 0.05]
    print('largest observer deviations', off)
    # noise band measured with the CPU oracle on two machines (scripts/qat_noise_probe.py): 38 of 181 ranges differ by more
    # than 5 %, 11 by more than 15 %, the worst by 33 % (single-element extremes behind 128-sample BatchNorms)
    assert np.median(dev) < 0.02 and np.mean(dev > 0.15) < 0.2 and dev.max() < 1.0, off
    w_rows = [i for i, k in enumerate(names['observers']) if k.endswith('weight_fake_quant')]
    np.testing.assert_allclose(got[w_rows][:, 2], want[w_rows][:, 2], rtol=2e-2)      # weight scales: nearly noise free
    gn_names = names['params']
    mine = dict(net.named_parameters())
    gn = np.array([mine[k.replace('.conv.bn.', '.bn.')].grad.double().norm().item() for k in gn_names])
    ratio = gn / np.maximum(g['g9_gradnorm'], 1e-12)
    big = g['g9_gradnorm'] > 1e-4 * g['g9_gradnorm'].max()
    worst = [(gn_names[i], float(ratio[i])) for i in np.argsort(-np.abs(np.log(np.maximum(ratio, 1e-30)))) if big[i]][:6]
    print('largest gradient-norm ratios', worst)
    # noise band, not a tolerance: the BatchNorm-bias gradients of layers 5 / 23 / 26 are the outliers of EVERY realisation (old
    # depthwise forward kernels: 1.81 / 0.46 / 0.40; walking forward kernel, same conv outputs bit for bit but another summation
    # grouping of the batch statistics: 3.08 / 0.23 / 0.48) -- one flipped rounding moves a batch statistic, which flips further
    # roundings downstream.  The tight check of the QAT arithmetic is test_qat_at_fixed_qparams_against_reference_g9b.
    lr = np.abs(np.log(ratio[big]))
    assert np.median(lr) < 0.1 and np.mean(lr > 1.0) <= 0.02 and np.all(lr < 2.0), worst
    net.eval()
    with torch.no_grad():
        ev = net(seeded_images(2, 256, 256, 7).to(DEV))[:, ::37]
    conf_err = np.abs(ev.cpu().numpy()[..., 4:] - g['g9_eval'][..., 4:])
    assert np.median(conf_err) < 5e-3 and np.mean(conf_err > 0.1) < 0.02


def test_qat_small_shape_first_step_reproduces_reference_and_observer_switch():
    """At 64x64 the first forward happens to meet no rounding-boundary disagreement with the reference run: every loss term
    agrees to 1e-5 (recorded when the fixtures were made; kept as a canary with a fallback to the noise band)."""
    net = _product()
    x, tgt = qat_batch(0, S=64)
    out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
    got = np.array([out[k].item() for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')])
    want = np.array([88.1607894897461, 4.020943641662598, 33.55781936645508, 50.58203125])      # reference, this container
    assert np.allclose(got, want, rtol=1e-4) or abs(got[0] - want[0]) <= 0.15 * want[0]
    out['loss'].mean().backward()
    before = net._qat_state.clone()
    tools.disable_observer(net)                              # trainer.py:299-300: ranges stay frozen from here on
    x, tgt = qat_batch(3, S=64)
    net(x.to(DEV), tuple(t.to(DEV) for t in tgt))['loss'].mean().backward()
    assert torch.equal(net._qat_state, before)
    opt = optim.Adam(net.parameters(), lr=1e-4)
    opt.step()
    assert all(torch.isfinite(p).all() for p in net.parameters())


@pytest.mark.parametrize('kind,shape', [('c1', (700, 32, 24)), ('c1', (513, 96, 200)), ('dw', (2, 9, 13, 32, 1)),
                                        ('dw', (1, 80, 72, 24, 2)), ('stem', (2, 16, 20, 32, 2))])
def test_min_max_planes_of_the_producers_and_the_folded_observer_against_the_oracle(kind, shape):
    """minmax != 0: the statistic rows of pq_conv1x1_fwd / pq_dw3x3_fwd (streaming and LDS-tiled) / pq_stem3x3_fwd carry the
    per-channel minimum and maximum of the outputs in planes 2 / 3; pq_bn_fwd_finalize(planes = 4, qstate) then leaves the
    observer state that oracle.qat.FakeQuant computes from a pass over act(bn(out)) -- twice (first batch, moving average)."""
    import torch.nn.functional as F
    l = abi.lib()
    g = torch.Generator().manual_seed(sum(shape))
    ACT = 1                                                    # ReLU (QAT nets are ReLU everywhere, model/parser.py:408-409)
    fq = oqat.FakeQuant(0, 255, False)
    state = torch.tensor([float('inf'), float('-inf'), 1.0, 0.0], device=DEV)
    mm = torch.tensor([-1, 0, 0, 0], dtype=torch.int32, device=DEV)
    for rep in range(2):
        if kind == 'c1':
            M, I, O = shape
            x = torch.randn(M, I, generator=g) * (1 + rep)
            w = torch.randn(O, I, generator=g) / I ** 0.5
            Cn, rows, ldo = O, M, (O + 3) // 4 * 4
            nblk = l.pq_conv1x1_blocks_k(M, O, I, 0)
            ldp = (O + 15) // 16 * 16
            part = torch.full((nblk * 4 * ldp,), float('nan'), device=DEV)
            out = torch.zeros(M, ldo, device=DEV)
            xd, wd = x.to(DEV), w.to(DEV)
            a = abi.Conv1x1Fwd(abi.Src(xd.data_ptr(), None, None, I, 0), wd.data_ptr(), None, out.data_ptr(), part.data_ptr(),
                               M, I, O, ldo, ldp, 1, 0, 0, 0, 0, 1)
            abi.check(l.pq_conv1x1_fwd(C.byref(a), _stream()))
        elif kind == 'dw':
            B, H, W, Cn, s = shape
            x = torch.randn(B, H, W, Cn, generator=g) * (1 + rep)
            w = torch.randn(Cn, 3, 3, generator=g) * 0.3
            OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
            rows, ldo = B * OH * OW, Cn
            nblk = l.pq_dw3x3_fwd_blocks(B, OH, OW, Cn, s)
            ldp = (Cn + 15) // 16 * 16
            part = torch.full((nblk * 4 * ldp,), float('nan'), device=DEV)
            out = torch.zeros(rows, ldo, device=DEV)
            xd, wd = x.to(DEV), w.to(DEV)
            a = abi.DwFwd(abi.Src(xd.data_ptr(), None, None, Cn, 0), wd.data_ptr(), out.data_ptr(), part.data_ptr(),
                          B, H, W, OH, OW, Cn, s, ldo, ldp, 1)
            abi.check(l.pq_dw3x3_fwd(C.byref(a), _stream()))
        else:
            B, H, W, Cn, s = shape
            x = torch.randn(B, 3, H, W, generator=g) * (1 + rep)
            w = torch.randn(Cn, 3, 3, 3, generator=g) * 0.3
            OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
            rows, ldo = B * OH * OW, Cn
            nblk = l.pq_ew_blocks(rows, Cn)
            ldp = (Cn + 15) // 16 * 16
            part = torch.full((nblk * 4 * ldp,), float('nan'), device=DEV)
            out = torch.zeros(rows, ldo, device=DEV)
            xd, wd = x.to(DEV), w.to(DEV)
            a = abi.StemFwd(xd.data_ptr(), wd.data_ptr(), out.data_ptr(), part.data_ptr(), B, 3, H, W, OH, OW, Cn, s, ldo, ldp, 1)
            abi.check(l.pq_stem3x3_fwd(C.byref(a), _stream()))
        torch.cuda.synchronize()
        raw = out[:, :Cn].cpu()
        pp = part.cpu().view(nblk, 4, ldp)
        assert torch.equal(pp[:, 2, :Cn].min(0).values, raw.min(0).values), 'plane 2: per-channel minimum'
        assert torch.equal(pp[:, 3, :Cn].max(0).values, raw.max(0).values), 'plane 3: per-channel maximum'
        gam, bet = (torch.rand(Cn, generator=g) + 0.5) * (1 - 2 * (torch.rand(Cn, generator=g) < 0.2).float()), torch.randn(Cn, generator=g) * 0.3
        gd, bd = gam.to(DEV), bet.to(DEV)
        rm, rv = torch.zeros(Cn, device=DEV), torch.ones(Cn, device=DEV)
        mean, inv, sc, sh = (torch.empty(Cn, device=DEV) for _ in range(4))
        f = abi.BnFwdFin(part.data_ptr(), gd.data_ptr(), bd.data_ptr(), rm.data_ptr(), rv.data_ptr(), None, mean.data_ptr(),
                         inv.data_ptr(), sc.data_ptr(), sh.data_ptr(), nblk, ldp, Cn, 1, float(rows), 1e-5, 0.1, None, None, None,
                         4, ACT, state.data_ptr(), mm.data_ptr(), 0.01)
        abi.check(l.pq_bn_fwd_finalize(C.byref(f), _stream()))
        torch.cuda.synchronize()
        # fmaf(scale, raw, shift) (one rounding: product and sum in fp64, then narrowed), then the activation
        post = torch.relu(torch.from_numpy(np.float32(sc.cpu().numpy()).astype(np.float64) * raw.numpy().astype(np.float64)
                                           + sh.cpu().numpy().astype(np.float64)).float())
        fq(post)
        got = state.cpu().tolist()
        want = [fq.min_val.item(), fq.max_val.item(), fq.scale.item(), float(fq.zero_point.item())]
        assert got == want, (rep, got, want)
        assert mm.cpu().tolist()[:3] == [-1, 0, 0], 're-armed'


def test_observer_folded_into_the_batchnorm_finalize_is_bit_identical_to_the_pass_over_the_tensor(monkeypatch):
    """Conv outputs that are never materialised in quantised form get their activation observer from per-channel min / max
    rows of the producer's epilogue + pq_bn_fwd_finalize (PQ_QAT_FOLD, default) instead of a pass over the tensor
    (pq_qact_fwd with out == NULL).  act(scale * x + shift) is monotone per channel, so both see exactly the same extremes:
    every observer state {min, max, scale, zero_point}, the losses and the gradients agree BIT FOR BIT over two steps."""
    runs = {}
    for fold in ('1', '0'):
        monkeypatch.setenv('PQ_QAT_FOLD', fold)
        net = _product()
        rec = []
        for step in range(2):
            x, tgt = qat_batch(step)
            net.zero_grad()
            out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
            out['loss'].mean().backward()
            rec.append((out['loss'].detach().clone(), net._qat_state.clone(), net._flat.grad.clone()))
        plan = list(net._plans.values())[0]
        n_obs = sum(1 for op in plan.fwd if op.code == abi.OP_QACT_FWD)
        n_fold = sum(1 for op in plan.fwd if op.code == abi.OP_BN_FWD_FIN and op.u.bnf.planes == 4)
        runs[fold] = (rec, n_obs, n_fold)
    (a, obs_a, fold_a), (b, obs_b, fold_b) = runs['1'], runs['0']
    assert fold_a >= 60 and fold_b == 0 and obs_b - obs_a == fold_a, (obs_a, fold_a, obs_b, fold_b)
    for (la, qa, ga), (lb, qb, gb) in zip(a, b):
        assert torch.equal(qa, qb), (qa - qb).abs().max()
        assert torch.equal(la, lb)
        assert torch.equal(ga, gb)


def test_fused_qat_plan_equals_the_materialising_plan(monkeypatch):
    """The default QAT plan (PQ_QAT_FUSE=1: consumers fake-quantise while loading, their gradient sinks apply the
    straight-through mask, nothing quantised is written for conv -> conv edges) -- the plan the config-5 benchmark times --
    against the materialising plan (PQ_QAT_FUSE=0, the one the teacher-forced G9b check replays): same arithmetic on the
    same values, only the place where fq() is applied moves.  Two train steps on two batches (the observers' moving averages
    move; no optimiser step in between: Adam turns the round-off of a mathematically zero gradient into a full-size update, so
    the second step would no longer start from the same weights): observer states and losses bit-identical; gradients equal up to fp32 summation order -- the BatchNorm-backward sums come out of
    different kernels' epilogues (consumer sinks instead of pq_qact_bwd), i.e. other row blocks and another partial order:
    measured 2e-6 norm-wise on the stem weight, asserted <= 1e-5 per parameter."""
    runs = {}
    for fuse in ('1', '0'):
        monkeypatch.setenv('PQ_QAT_FUSE', fuse)
        net = _product()
        rec = []
        for step in range(2):
            x, tgt = qat_batch(step)
            net.zero_grad()
            out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
            out['loss'].mean().backward()
            rec.append((torch.stack([out[k] for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')]).detach().clone(),
                        net._qat_state.clone(), net._flat.grad.clone()))
        plan = list(net._plans.values())[0]
        runs[fuse] = (rec, sum(1 for t in plan.tensors if t is not None and getattr(t, 'fuse_inner', None) is not None), net)
    (a, fused_a, net_a), (b, fused_b, _) = runs['1'], runs['0']
    assert fused_a >= 60 and fused_b == 0, (fused_a, fused_b)
    for step, ((la, qa, ga), (lb, qb, gb)) in enumerate(zip(a, b)):
        assert torch.equal(qa, qb), (step, (qa - qb).abs().max())
        assert torch.equal(la, lb), (step, la, lb)
        if not torch.equal(ga, gb):
            flat = net_a._flat
            gmax = max(gb[o:o + p.numel()].double().norm().item() for p, o in zip(flat.params, flat.offsets))
            for p, o in zip(flat.params, flat.offsets):
                da, db = ga[o:o + p.numel()].double(), gb[o:o + p.numel()].double()
                # absolute floor: the bias of a BatchNorm whose only consumer is a linear conv + train-mode BatchNorm has a
                # mathematically ZERO gradient (the next BatchNorm removes the shift); what both plans produce there is the
                # round-off of a 10^6-term cancelling sum (~1e-3 against gradient norms of ~5e4), not a value
                assert (da - db).norm().item() <= 1e-5 * db.norm().item() + 1e-7 * gmax, (step, o, (da - db).norm().item(), db.norm().item(), gmax)


def test_two_stream_forward_equals_the_single_stream_forward(monkeypatch):
    """PQ_FWD_SIDE=0 (everything on one stream) against the default two-stream forward (detection-head branches and the loss
    kernels beside the next level's neck): losses, running statistics and every gradient bit-identical, float and QAT plans.
    A race between the streams would show up here as a difference.  (The BatchNorm statistics hand-over is switched
    off: it only applies to main-stream layers, so the two plans would round invstd differently in the head branches.)"""
    monkeypatch.setenv('PQ_BN_ATOMIC', '0')
    for qat_plan in (False, True):
        runs = {}
        for side in ('1', '0'):
            monkeypatch.setenv('PQ_FWD_SIDE', side)
            if qat_plan:
                net = _product()
            else:
                net = DetectionModel(io.StringIO(zoo.cfg_text('mobilenetv2-fpn')))
                seeded_fill(net, 0)
                net = net.to(DEV)
                net.train()
            rec = []
            for step in range(2):
                x, tgt = qat_batch(step)
                net.zero_grad()
                out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
                out['loss'].mean().backward()
                rec.append((out['loss'].detach().clone(), net._flat.grad.clone(),
                            torch.cat([b.detach().flatten().float() for b in net.buffers()])))
            plan = list(net._plans.values())[0]
            runs[side] = (rec, bool(getattr(plan, 'fwd_two_streams', False)))
            # the two-stream float plan also moves the weight transposes of the backward into the forward's side stream and
            # queues the side heads' loss kernels without a fork event (PQ_OPF_NOFORK): both are part of what is compared
            nofork = [op.code for op in plan.fwd if op.flags & abi.OPF_NOFORK]
            if side == '1':
                assert nofork.count(abi.OP_LOSS_FWD) == 2 and plan.tr_in_forward == (not qat_plan)
                assert (abi.OP_TRANSPOSE in nofork) == (not qat_plan)
            else:
                assert not nofork and not plan.tr_in_forward
        assert runs['1'][1] and not runs['0'][1]
        for (la, ga, ba), (lb, gb, bb) in zip(runs['1'][0], runs['0'][0]):
            assert torch.equal(la, lb) and torch.equal(ga, gb) and torch.equal(ba, bb), qat_plan


def test_config5_full_size_properties():
    """BASELINE config 5 at its stated shape (mobilenetv2-fpn QAT, 512 x 512, batch 12, yamls/quant.yaml:16): the step is
    finite, a replay from the same state is bit-identical (losses, observer table, gradients), and after
    tools.disable_observer (trainer.py:299-300) the observers stay frozen while the fake-quantised step keeps training."""
    from pqdet_amd.dataset import labels
    net = _product()
    x = torch.randn(12, 3, 512, 512, generator=torch.Generator().manual_seed(0)).to(DEV)
    tgt = tuple(t.to(DEV) for t in labels.synthetic_targets(12, 512, 20, seed=1))
    sd0 = {k: v.clone() for k, v in net.plain_state_dict().items()}
    runs = []
    for _ in range(2):
        nn_load = torch.nn.Module.load_state_dict(net, sd0)
        assert not nn_load.missing_keys and not nn_load.unexpected_keys
        net._plans.clear()
        net.zero_grad()
        out = net(x, tgt)
        out['loss'].mean().backward()
        net.check_nan()
        runs.append((out['loss'].detach().clone(), net._qat_state.clone(), net._flat.grad.clone()))
    assert all(torch.isfinite(t).all() for t in runs[0])
    assert all(torch.equal(a, b) for a, b in zip(runs[0], runs[1]))
    assert runs[0][2].abs().max().item() > 0
    moved = (runs[0][1] != sd0['_qat_state']).any(1).float().mean().item()
    assert moved > 0.9, moved                                   # observers on: (almost) every row was updated
    tools.disable_observer(net)
    frozen = net._qat_state.clone()
    net.zero_grad()
    out = net(x, tgt)
    out['loss'].mean().backward()
    net.check_nan()
    assert torch.equal(net._qat_state, frozen)
    assert torch.isfinite(out['loss']).all() and torch.isfinite(net._flat.grad).all()


# ------------------------------------------------------------------------------------------------- 4. fixed qparams (G9b)
def _load_g9b_state(net, golden_r2, r2_names):
    """BatchNorm running statistics + every observer's qparams as recorded from the reference before the G9b pass."""
    own = net.plain_state_dict()
    flat = torch.from_numpy(golden_r2['g9b_bn_stats'])
    off = 0
    want_obs = dict(zip(r2_names['g9b_observers'], golden_r2['g9b_observers']))
    with torch.no_grad():
        for k in r2_names['g9b_bn_keys']:                       # 'module_list.i.conv.bn.running_*' in the reference's fused layout
            t = own[k.replace('.conv.bn.', '.bn.')]
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        assert off == flat.numel()
        assert set(net.qat_observers()) == set(want_obs)
        for key, row in net._qat_index.items():
            if key[0] == 'in':
                name = 'qstub.activation_post_process'
            elif key[0] == 'w':
                name = 'module_list.%d.conv.weight_fake_quant' % key[1]
            else:
                kind = 'conv' if net.module_list[key[1]]._type == 'convolutional' else 'ffunc'
                name = 'module_list.%d.%s.activation_post_process' % (key[1], kind)
            net._qat_state[row] = torch.tensor(want_obs[name], dtype=torch.float32)
    tools.disable_observer(net)
    net._plans.clear()
    return want_obs


def test_qat_at_fixed_qparams_against_reference_g9b(golden_r2, r2_names, monkeypatch):
    """G9b: the reference QAT model after two passes, observers DISABLED (trainer.py:299-300), one more train-mode pass; the
    HIP model is loaded with exactly that state (BatchNorm running statistics + every observer's qparams; weights by the
    seeded fill).

    (a) Layer by layer, teacher-forced: every HIP layer is fed the reference's quantised input (the oracle's, which equals the
        reference's bit for bit) and must reproduce its quantised output -- identical values except where a pre-rounding value
        sits on a rounding boundary (<= 1e-3 of the elements), and those by exactly ONE quantisation step.
    (b) End to end the same network is a chaotic map: one flipped rounding in layer k flips ~3 in layer k+1 (measured:
        0, 5e-5, 1.4e-4, 4e-4, 3e-3 ... of the values moved after layers 1..5, scripts/qat_fixed_hip_probe.py), and the
        reference implementation itself does not reproduce its own loss across two CPUs to better than 3.6e-3 (the oracle on
        the GPU box's host vs the fixture).  So end to end the assertion is: the first layers agree as in (a), and the losses
        stay within 5x the oracle's own cross-machine deviation on this very box (never below north_star's 1e-3)."""
    monkeypatch.setenv('PQ_QAT_FUSE', '0')             # materialise the quantised tensors so that they can be read back / fed
    from test_golden_r2 import _targets
    net = _product(0)
    want_obs = _load_g9b_state(net, golden_r2, r2_names)
    ref = _oracle_at_g9b(golden_r2, r2_names)
    taps = {}
    ref.input_fq.register_forward_hook(lambda mod, inp, o: taps.__setitem__(-1, o.detach()))
    for i in range(len(ref.module_list)):
        if isinstance(ref.a_fq[i], oqat.FakeQuant):
            ref.a_fq[i].register_forward_hook(lambda mod, inp, o, i=i: taps.__setitem__(i, o.detach()))
    x, tgt = seeded_images(2, 256, 256, 2), _targets(2, 256, 20, 3)
    want = ref(x, tgt)
    out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
    torch.cuda.synchronize()
    plan = [p for p in net._plans.values() if p.with_loss][-1]

    def step_of(i):
        kind = 'conv' if net.module_list[i]._type == 'convolutional' else 'ffunc'
        return float(want_obs['module_list.%d.%s.activation_post_process' % (i, kind)][2])

    def nhwc(t):
        return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])

    def moved(i):
        t = plan.tensors[i]
        d = (t.raw[:, :t.C].cpu() - nhwc(taps[i])).abs() / step_of(i)
        assert (d - d.round()).abs().max().item() < 2e-2, (i, 'off the quantisation grid')
        return (d > 0.5).float().mean().item(), d.max().item()

    # ---- (b) free-running: the first layers, then the losses
    free = {i: moved(i) for i in (0, 1, 2)}
    assert all(f <= 1e-3 and m <= 1.01 for f, m in free.values()), free
    # the five layers stored in the fixture are the oracle's, bit for bit, wherever this test runs on the recording machine;
    # elsewhere the oracle drifts like any second evaluation: compare the fixture with the oracle at layer 1 only
    w1 = torch.from_numpy(golden_r2['g9b_act1'])
    s1 = int(golden_r2['g9b_act1_stride'][0])
    assert ((nhwc(taps[1]).reshape(-1)[::s1] - w1).abs() > 0.5 * step_of(1)).float().mean().item() <= 1e-3
    ref_loss = golden_r2['g9b_loss'][0]
    band = max(1e-3, 5.0 * abs(want['loss'].item() - ref_loss) / ref_loss)
    assert abs(out['loss'].item() - ref_loss) / ref_loss <= band, (out['loss'].item(), want['loss'].item(), ref_loss)

    # ---- (a) teacher-forced, every conv layer that reads one materialised tensor
    l = abi.lib()
    ops_of = {}
    for k, li in enumerate(plan.fwd_layer):
        ops_of.setdefault(li, []).append(k)
    worst = {}
    mods = net.module_list
    for i in range(1, len(mods)):
        if mods[i]._type != 'convolutional' or i not in taps or (i - 1) not in taps or mods[i - 1]._type == 'yolo':
            continue
        src = plan.tensors[i - 1]
        if src.raw.shape[0] != taps[i - 1].shape[0] * taps[i - 1].shape[2] * taps[i - 1].shape[3]:
            continue
        src.raw[:, :src.C].copy_(nhwc(taps[i - 1]).to(DEV))                  # the reference's quantised input
        idx = ops_of[i]
        sub = (abi.Op * len(idx))(*[plan.fwd_arr[k] for k in idx])
        abi.check(l.pq_run_ops(sub, len(idx), _stream()), 'replay of layer %d' % i)
        torch.cuda.synchronize()
        worst[i] = moved(i)
    assert len(worst) >= 60, len(worst)
    bad = {i: v for i, v in worst.items() if v[0] > 1e-3 or v[1] > 1.01}
    assert not bad, bad
    print('teacher-forced layers: %d, worst moved fraction %.2e' % (len(worst), max(v[0] for v in worst.values())))
